"""Generator of ominix-mlx_amd/csrc/attn_flash4_body.inc: the per-unit body of the 4-wave flash-attention kernel (attn_flash4.hip) as ONE
inline-asm template with hand-allocated registers and hand-placed fillers per MFMA gap.

Why generated assembly: written in HIP, hipcc put the S accumulators in AGPRs (~150 v_accvgpr_read per key tile), spilled ~100 registers
and reloaded the LDS-DMA offsets from scratch behind vmcnt(0) in every MFMA gap (EXPERIMENTS.md R5-1).  Here the register FILES are ours:

  AGPR  a[0:127]    O^T accumulators  O[qb][db] = a[(qb*4+db)*16 ..+15]
        a[128:191]  Q fragments       Q[qb][i]  = a[128 + (qb*8+i)*4 ..+3]      (B operand of S^T = K Q^T)
        a[192:255]  K fragments       K[kb][i]  = a[192 + (kb*8+i)*4 ..+3]      (A operand; read from LDS one phase ahead)
  VGPR  v[0:31]     left to hipcc (the 24 vector operands of the statement live here)
        v[32:95]    S buffer 0  S[kb][qb] = 16 registers each     v[96:159]  S buffer 1
        v[160:191]  P (bf16 pairs) P[j][qb] = 4 registers each
        v[192:223]  V^T fragments of even / odd 16-key steps (4 fragments x 4 registers each)
        v[224:255]  softmax state and temporaries
  SGPR  s[60:95]    ours (declared clobbered); everything else through operands

Operand order of the statement (all inputs): see OPERANDS below; attn_flash4.hip lists them in the same order.

Tile step T (ring slot T & 3 is static: the loop is unrolled by 4, Tk % 256 == 0), one wave per SIMD:
  phase A: 32 MFMAs S_next = K_{T+1} Q^T     | p = 2^t in place, row sums, cvt to bf16 P | V_T fragments of key step 0
  phase B: s_waitcnt vmcnt(8); s_barrier
           32 MFMAs O += V_T^T P             | LDS-DMA of V_{T+3}, K_{T+5} | V_T fragments of steps 1..3 | K_{T+2} fragments |
                                               row max of S_next, deferred-rescale decision, t = s c - m
Hazards hipcc would pad and we keep apart by construction: an accumulator recurs every 4th (S) / 8th (O) MFMA; S is first read by VALU
two MFMA gaps and a barrier after its last MFMA; P is written a phase before its MFMA; VALU touches O only after explicit s_nops.
"""
import os
import sys

THR = 8.0          # deferred-rescale threshold, base-2 exponent units (the kernel's second instantiation uses 0)

OPERANDS = [
    # (name, constraint)  -- vector operands first
    ("koff0", "v"), ("koff1", "v"), ("koff2", "v"), ("koff3", "v"),
    ("voff0", "v"), ("voff1", "v"), ("voff2", "v"), ("voff3", "v"),
    ("kro0", "v"), ("kro1", "v"), ("kro2", "v"), ("kro3", "v"), ("kro4", "v"), ("kro5", "v"), ("kro6", "v"), ("kro7", "v"),
    ("vro0", "v"), ("vro1", "v"), ("vro2", "v"), ("vro3", "v"),
    ("qoff0", "v"), ("qoff1", "v"), ("ooff0", "v"), ("ooff1", "v"),
    ("qbase", "s"), ("obase", "s"), ("kcur", "s"), ("vcur", "s"), ("knext", "s"), ("vnext", "s"),
    ("stride", "s"), ("nt", "s"), ("c2", "s"), ("kdst", "s"), ("vdst", "s"),
    ("rowmask0", "s"), ("rowmask1", "s"),
]
OP = {name: "%%%d" % i for i, (name, _) in enumerate(OPERANDS)}

# ---- register map ----
def S(buf, kb, qb):
    return 32 + buf * 64 + (kb * 2 + qb) * 16
def P(j, qb):
    return 160 + (j * 2 + qb) * 4
def VF(par, db):
    return 192 + par * 16 + db * 4
MX = [224, 225]
MR = [226, 227]
MN = [228, 229]
L = [[230, 231], [232, 233]]     # two partial row sums per query block
AL = [234, 235]
TMP = list(range(236, 256))      # 20 temporaries
def O(qb, db):
    return (qb * 4 + db) * 16
def Q(qb, i):
    return 128 + (qb * 8 + i) * 4
def KF(kb, i):
    return 192 + (kb * 8 + i) * 4
def vr(lo, n=1):
    return "v%d" % lo if n == 1 else "v[%d:%d]" % (lo, lo + n - 1)
def ar(lo, n=1):
    return "a%d" % lo if n == 1 else "a[%d:%d]" % (lo, lo + n - 1)

# our scalar registers
S_KPTR, S_VPTR = 60, 62          # pairs
S_KT, S_VT = 64, 65
S_NEED = [66, 68]                # pairs
S_T64 = 70                       # pair
S_LOOP = 72
S_M0 = 73
S_EXEC = 74                      # pair
S_RET = 78                       # pair
S_TMP = 80
def sp(lo):
    return "s[%d:%d]" % (lo, lo + 1)


class Gen:
    def __init__(self, thr, diag=()):
        self.diag = set(diag)     # timing-only builds (results garbage): "nodma", "novalu", "nolds" drop that class of instructions
        self.lines = []
        self.lds = []            # outstanding LDS operations (ids) in issue order
        self.next_id = 0
        self.thr = thr
        self.uid = 0

    def e(self, s):
        op = s.split()[0]
        if "nodma" in self.diag and (op == "global_load_lds_dwordx4" or s.startswith("s_add_u32 m0")):
            return
        if "novalu" in self.diag and op in ("v_exp_f32", "v_add_f32", "v_cvt_pk_bf16_f32", "v_max_f32", "v_max3_f32", "v_fma_f32", "v_mul_f32",
                                            "v_permlane32_swap_b32", "v_cmp_gt_f32_e64", "v_cndmask_b32_e64", "v_mov_b32", "s_getpc_b64", "s_branch") \
                and "F4_store" not in s:
            return
        if "novalu" in self.diag and (s.startswith("s_cbranch_scc0 F4_norescale") or s.startswith("s_add_u32 s%d, s%d, 12" % (S_RET, S_RET))):
            return
        if "expmov" in self.diag and op == "v_exp_f32":
            s = s.replace("v_exp_f32", "v_mov_b32")
        if "nobar" in self.diag and (op == "s_barrier" or s.startswith("s_waitcnt vmcnt(8)")):
            return
        if "noadd" in self.diag and op == "v_add_f32":
            return
        if "nolds" in self.diag and (op.startswith("ds_read") or s.startswith("s_waitcnt lgkmcnt")):
            return
        self.lines.append(s)

    def lds_op(self, s):
        self.e(s)
        self.lds.append(self.next_id)
        self.next_id += 1
        return self.next_id - 1

    def lds_wait(self, ids):
        """wait until every operation in ids has returned (LDS operations return in order)"""
        ids = [i for i in ids if i in self.lds]
        if not ids:
            return
        newest = max(self.lds.index(i) for i in ids)
        n_after = min(15, len(self.lds) - 1 - newest)
        self.e("s_waitcnt lgkmcnt(%d)" % n_after)
        self.lds = self.lds[len(self.lds) - n_after:] if n_after else []

    def lds_wait_all(self):
        self.e("s_waitcnt lgkmcnt(0)")
        self.lds = []

    def label(self, stem):
        self.uid += 1
        return "F4_%s_%d_%%=" % (stem, self.uid)


def mfma_s(g_, buf, kb, qb, i):
    acc = vr(S(buf, kb, qb), 16)
    g_.e("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, ar(KF(kb, i), 4), ar(Q(qb, i), 4), "0" if i == 0 else acc))

def mfma_o(g_, qb, db, par, j):
    acc = ar(O(qb, db), 16)
    g_.e("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (acc, vr(VF(par, db), 4), vr(P(j, qb), 4), acc))

def read_v(g_, slot, j, db, half):
    """one transposing read: half 0 -> registers 0..1 of the fragment (keys 4 hi + 0..3), half 1 -> registers 2..3 (keys 8 + 4 hi + 0..3)"""
    dst = VF(j & 1, db) + 2 * half
    return g_.lds_op("ds_read_b64_tr_b16 %s, %s offset:%d" % (vr(dst, 2), OP["vro%d" % db], slot * 16384 + j * 4096 + half * 2048))

def read_k(g_, slot, kb, i):
    return g_.lds_op("ds_read_b128 %s, %s offset:%d" % (ar(KF(kb, i), 4), OP["kro%d" % i], slot * 16384 + kb * 8192))

def elem(n):
    """softmax element n = 0..63 in (key step j, query block, e) order -> (register offset inside the S buffer, j, qb, e)"""
    j, qb, e_ = n >> 4, (n >> 3) & 1, n & 7
    return (S(0, j >> 1, qb) - 32) + 8 * (j & 1) + e_, j, qb, e_


def phase_a(g_, slot, cur, nxt, has_next):
    """p = 2^t (in place in S[cur]), row sums, P; S[nxt] = K Q^T; the transposing reads of V_T key step 0 (gaps 0..7) and the first half of
    key step 1 (gaps 28..31)"""
    g_.lds_wait_all()                       # K fragments (and everything else phase B read) are in
    pending = []
    vids = {}
    for gap in range(32):
        if has_next:
            i, kb, qb = gap >> 2, (gap >> 1) & 1, gap & 1
            mfma_s(g_, nxt, kb, qb, i)
        regs = []
        for u in range(2):
            off, j, qb, e_ = elem(2 * gap + u)
            r = 32 + cur * 64 + off
            g_.e("v_exp_f32 %s, %s" % (vr(r), vr(r)))
            regs.append((r, j, qb, e_))
        for op in pending:
            g_.e(op)
        (r0, j, qb, e0), (r1, _, _, _) = regs
        pending = ["v_add_f32 %s, %s, %s" % (vr(L[qb][0]), vr(L[qb][0]), vr(r0)),
                   "v_add_f32 %s, %s, %s" % (vr(L[qb][1]), vr(L[qb][1]), vr(r1)),
                   "v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(P(j, qb) + (e0 >> 1)), vr(r0), vr(r1))]
        if gap < 8:
            vids[(0, gap >> 1, gap & 1)] = read_v(g_, slot, 0, gap >> 1, gap & 1)
        if gap >= 28:
            k = gap - 28
            vids[(1, k >> 1, k & 1)] = read_v(g_, slot, 1, k >> 1, k & 1)
    for op in pending:
        g_.e(op)
    return vids


def phase_b(g_, slot, nxt, has_next, vids, in_tail):
    """O += V^T P with the DMA, the fragment reads and the start of the next tile's softmax in the gaps.  The stream cursors cross into the
    next unit at static places (nt % 4 == 0): V (tile + 3 issued here, tile + 4 next) after the first tile of the unit's LAST group, K
    (tile + 5 here, tile + 6 next) after the third tile of the group before it = the loop's last iteration"""
    g_.e("s_waitcnt vmcnt(8)")
    g_.e("s_barrier")
    fill = {gap: [] for gap in range(32)}
    # ---- LDS-DMA: V_{T+3} pieces in gaps 0..3, K_{T+5} in gaps 4..7 (m0 write first in its gap, the DMA last: an instruction in between) ----
    vslot, kslot = (slot + 3) & 3, (slot + 1) & 3
    for it in range(4):
        fill[it].insert(0, ("salu", "s_add_u32 m0, %s, %d" % (OP["vdst"], vslot * 16384 + it * 4096)))
        fill[it].append(("dma", "global_load_lds_dwordx4 %s, %s" % (OP["voff%d" % it], sp(S_VPTR))))
    if in_tail and slot == 0:
        vcur = [("salu", "s_mov_b64 %s, %s" % (sp(S_VPTR), OP["vnext"]))]
    else:
        vcur = [("salu", "s_add_u32 s%d, s%d, %s" % (S_VPTR, S_VPTR, OP["stride"])), ("salu", "s_addc_u32 s%d, s%d, 0" % (S_VPTR + 1, S_VPTR + 1))]
    fill[8] += vcur
    if has_next:
        for it in range(4):
            fill[4 + it].insert(0, ("salu", "s_add_u32 m0, %s, %d" % (OP["kdst"], kslot * 16384 + it * 4096)))
            fill[4 + it].append(("dma", "global_load_lds_dwordx4 %s, %s" % (OP["koff%d" % it], sp(S_KPTR))))
        kcur = [("salu", "s_add_u32 s%d, s%d, %s" % (S_KPTR, S_KPTR, OP["stride"])), ("salu", "s_addc_u32 s%d, s%d, 0" % (S_KPTR + 1, S_KPTR + 1))]
        if slot == 2 and not in_tail:
            kcur += [("salu", "s_cmp_eq_u32 s%d, 1" % S_LOOP), ("salu", "s_cselect_b64 %s, %s, %s" % (sp(S_KPTR), OP["knext"], sp(S_KPTR)))]
        fill[9] += kcur
    # ---- V^T fragments: second half of key step 1 in gaps 0..3, step 2 in gaps 4..11, step 3 in gaps 12..19 (each step complete five gaps
    #      before its first MFMA: ONE wait per step) ----
    for gap in range(20):
        k = gap + 4
        j, db, half = (k >> 3) + 1, (k & 7) >> 1, k & 1
        fill[gap].append(("vread", (j, db, half)))
    if has_next:
        # ---- K_{T+2} fragments: one per gap over gaps 8..23 ----
        for gap in range(8, 24):
            f = gap - 8
            fill[gap].append(("kread", (f & 1, f >> 1)))          # (kb, i): head-dim step outermost, the order phase A consumes them in
        # ---- row max of S_next: gaps 1..16, one v_max3 per query block per gap ----
        for k in range(16):
            kb, r = k >> 3, 2 * (k & 7)
            for q2 in range(2):
                a0 = S(nxt, kb, q2) + r
                if k == 0:
                    fill[1 + k].append(("valu", "v_max_f32 %s, %s, %s" % (vr(MX[q2]), vr(a0), vr(a0 + 1))))
                else:
                    fill[1 + k].append(("valu", "v_max3_f32 %s, %s, %s, %s" % (vr(MX[q2]), vr(MX[q2]), vr(a0), vr(a0 + 1))))
        # ---- both halves of a row, the deferred-rescale decision: gaps 17..20 ----
        t0, t1, t2, t3 = TMP[0], TMP[1], TMP[2], TMP[3]
        fill[17] += [("early", "v_mov_b32 %s, %s" % (vr(t0), vr(MX[0]))), ("early", "v_mov_b32 %s, %s" % (vr(t1), vr(MX[1])))]
        # (two instructions -- this gap's fragment reads -- separate the copies from the swaps)
        fill[17] += [("late", "v_permlane32_swap_b32 %s, %s" % (vr(t0), vr(MX[0]))), ("late", "v_permlane32_swap_b32 %s, %s" % (vr(t1), vr(MX[1])))]
        fill[18] += [("valu", "v_max_f32 %s, %s, %s" % (vr(MX[0]), vr(MX[0]), vr(t0))), ("valu", "v_max_f32 %s, %s, %s" % (vr(MX[1]), vr(MX[1]), vr(t1)))]
        for q2 in range(2):
            fill[18].append(("late", "v_mul_f32 %s, %s, %s" % (vr(MX[q2]), OP["c2"], vr(MX[q2]))))
        for q2, tt in ((0, t2), (1, t3)):
            if g_.thr > 0:
                fill[19].append(("valu", "v_add_f32 %s, 0x%08x, %s" % (vr(tt), f32_bits(g_.thr), vr(MR[q2]))))
                fill[19].append(("late", "v_cmp_gt_f32_e64 %s, %s, %s" % (sp(S_NEED[q2]), vr(MX[q2]), vr(tt))))
            else:
                fill[19].append(("late", "v_cmp_gt_f32_e64 %s, %s, %s" % (sp(S_NEED[q2]), vr(MX[q2]), vr(MR[q2]))))
            fill[20].append(("valu", "v_cndmask_b32_e64 %s, %s, %s, %s" % (vr(MN[q2]), vr(MR[q2]), vr(MX[q2]), sp(S_NEED[q2]))))
        # ---- t = s c - m: 64 values: 3 in gap 20, 5 per gap in 21..23, the rest over 24..31 ----
        plan = [(20, 3), (21, 5), (22, 5), (23, 5)]
        left = 64 - sum(n for _, n in plan)
        for k, gap in enumerate(range(24, 32)):
            plan.append((gap, left // 8 + (1 if k < left % 8 else 0)))
        x = 0
        for gap, n in plan:
            for _ in range(n):
                kb, q2, r = x >> 5, (x >> 4) & 1, x & 15
                reg = S(nxt, kb, q2) + r
                fill[gap].append(("late" if gap == 20 else "valu", "v_fma_f32 %s, %s, %s, -%s" % (vr(reg), vr(reg), OP["c2"], vr(MN[q2]))))
                x += 1
        assert x == 64
    # ---- emit ----
    for gap in range(32):
        j, db, qb = gap >> 3, (gap >> 1) & 3, gap & 1
        if gap % 8 == 0:
            g_.lds_wait([vids[(j, d, h)] for d in range(4) for h in range(2)])
        mfma_o(g_, qb, db, j & 1, j)
        order = [f for f in fill[gap] if f[0] == "early"] + [f for f in fill[gap] if f[0] not in ("dma", "late", "early")] + \
                [f for f in fill[gap] if f[0] == "late"] + [f for f in fill[gap] if f[0] == "dma"]
        for kind, what in order:
            if kind == "vread":
                jj, dd, hh = what
                vids[(jj, dd, hh)] = read_v(g_, slot, jj, dd, hh)
            elif kind == "kread":
                kb, i = what
                read_k(g_, (slot + 2) & 3, kb, i)
            else:
                g_.e(what)
    if has_next:
        # the rare half of the deferred rescale, as a call into the shared block
        skip = g_.label("norescale")
        g_.e("s_or_b64 %s, %s, %s" % (sp(S_T64), sp(S_NEED[0]), sp(S_NEED[1])))
        g_.e("s_cmp_lg_u64 %s, 0" % sp(S_T64))
        g_.e("s_cbranch_scc0 %s" % skip)
        g_.e("s_getpc_b64 %s" % sp(S_RET))
        g_.e("s_add_u32 s%d, s%d, 12" % (S_RET, S_RET))
        g_.e("s_addc_u32 s%d, s%d, 0" % (S_RET + 1, S_RET + 1))
        g_.e("s_branch F4_rescale_%=")
        g_.e("%s:" % skip)


def f32_bits(x):
    import struct
    return struct.unpack("<I", struct.pack("<f", x))[0]


def tile_step(g_, slot, cur, nxt, has_next, in_tail):
    vids = phase_a(g_, slot, cur, nxt, has_next)
    phase_b(g_, slot, nxt, has_next, vids, in_tail)


def drain(g_):
    g_.e("s_nop 7")
    g_.e("s_nop 7")
    g_.e("s_nop 7")


def generate(thr, diag=()):
    g_ = Gen(thr, diag)
    e = g_.e
    e("s_mov_b32 s%d, m0" % S_M0)
    e("s_mov_b64 %s, exec" % sp(S_EXEC))
    # ---- stream cursors: K at tile 4 of this unit (issued below), V at tile 3 (issued by tile 0's phase B) ----
    e("s_mov_b64 %s, %s" % (sp(S_KPTR), OP["kcur"]))
    e("s_mov_b64 %s, %s" % (sp(S_VPTR), OP["vcur"]))
    e("s_lshl_b32 s%d, %s, 2" % (S_TMP, OP["stride"]))
    e("s_add_u32 s%d, s%d, s%d" % (S_KPTR, S_KPTR, S_TMP))
    e("s_addc_u32 s%d, s%d, 0" % (S_KPTR + 1, S_KPTR + 1))
    e("s_mul_i32 s%d, %s, 3" % (S_TMP, OP["stride"]))
    e("s_add_u32 s%d, s%d, s%d" % (S_VPTR, S_VPTR, S_TMP))
    e("s_addc_u32 s%d, s%d, 0" % (S_VPTR + 1, S_VPTR + 1))
    # ---- Q fragments straight into AGPRs; K_T fragments (slot 0); O = 0; row sums = 0 ----
    for qb in range(2):
        for i in range(8):
            e("global_load_dwordx4 %s, %s, %s offset:%d" % (ar(Q(qb, i), 4), OP["qoff%d" % qb], OP["qbase"], i * 32))
    for i in range(8):
        for kb in range(2):
            read_k(g_, 0, kb, i)
    for r in range(128):
        e("v_accvgpr_write_b32 a%d, 0" % r)
    for qb in range(2):
        for u in range(2):
            e("v_mov_b32 %s, 0" % vr(L[qb][u]))
    e("s_waitcnt vmcnt(0)")
    g_.lds_wait_all()
    # ---- S_T alone ----
    for i in range(8):
        for kb in range(2):
            for qb in range(2):
                mfma_s(g_, 0, kb, qb, i)
    # every wave has read K_T: its slot (0) takes K_{T+4}
    e("s_barrier")
    for it in range(4):
        e("s_add_u32 m0, %s, %d" % (OP["kdst"], it * 4096))
        e("s_nop 0")
        e("global_load_lds_dwordx4 %s, %s" % (OP["koff%d" % it], sp(S_KPTR)))
    e("s_add_u32 s%d, s%d, %s" % (S_KPTR, S_KPTR, OP["stride"]))
    e("s_addc_u32 s%d, s%d, 0" % (S_KPTR + 1, S_KPTR + 1))
    drain(g_)
    # K_{T+1} fragments (slot 1) for the first phase A
    for i in range(8):
        for kb in range(2):
            read_k(g_, 1, kb, i)
    # first tile's row max sets m; t = s c - m
    for q2 in range(2):
        for k in range(16):
            kb, r = k >> 3, 2 * (k & 7)
            a0 = S(0, kb, q2) + r
            if k == 0:
                e("v_max_f32 %s, %s, %s" % (vr(MX[q2]), vr(a0), vr(a0 + 1)))
            else:
                e("v_max3_f32 %s, %s, %s, %s" % (vr(MX[q2]), vr(MX[q2]), vr(a0), vr(a0 + 1)))
    t0, t1 = TMP[0], TMP[1]
    e("v_mov_b32 %s, %s" % (vr(t0), vr(MX[0])))
    e("v_mov_b32 %s, %s" % (vr(t1), vr(MX[1])))
    e("s_nop 1")
    e("v_permlane32_swap_b32 %s, %s" % (vr(t0), vr(MX[0])))
    e("v_permlane32_swap_b32 %s, %s" % (vr(t1), vr(MX[1])))
    for q2, tt in ((0, t0), (1, t1)):
        e("v_max_f32 %s, %s, %s" % (vr(MX[q2]), vr(MX[q2]), vr(tt)))
        e("v_mul_f32 %s, %s, %s" % (vr(MR[q2]), OP["c2"], vr(MX[q2])))
    for x in range(64):
        kb, q2, r = x >> 5, (x >> 4) & 1, x & 15
        reg = S(0, kb, q2) + r
        e("v_fma_f32 %s, %s, %s, -%s" % (vr(reg), vr(reg), OP["c2"], vr(MR[q2])))
    # ---- the tiles: groups of four (ring slot = position in the group, S buffer = its parity); the unit's last tile has no successor ----
    loop, tail = "F4_loop_%=", "F4_tail_%="
    e("s_lshr_b32 s%d, %s, 2" % (S_LOOP, OP["nt"]))
    e("s_sub_u32 s%d, s%d, 1" % (S_LOOP, S_LOOP))
    e("s_cmp_eq_u32 s%d, 0" % S_LOOP)
    e("s_cbranch_scc1 %s" % tail)
    e("%s:" % loop)
    for pos in range(4):
        tile_step(g_, pos, pos & 1, (pos + 1) & 1, True, False)
    e("s_sub_u32 s%d, s%d, 1" % (S_LOOP, S_LOOP))
    e("s_cmp_lg_u32 s%d, 0" % S_LOOP)
    e("s_cbranch_scc1 %s" % loop)
    e("%s:" % tail)
    for pos in range(4):
        tile_step(g_, pos, pos & 1, (pos + 1) & 1, pos < 3, True)
    # ---- normalise and store: lane holds out[row][32 db + 8 g + 4 hi + 0..3] ----
    drain(g_)
    e("s_branch F4_store_%=")
    # ---- shared block: the rare half of the deferred rescale (O, l at the old scale; m moves) ----
    e("F4_rescale_%=:")
    drain(g_)
    for q2 in range(2):
        e("v_sub_f32 %s, %s, %s" % (vr(AL[q2]), vr(MR[q2]), vr(MN[q2])))
        e("v_exp_f32 %s, %s" % (vr(AL[q2]), vr(AL[q2])))
        e("v_mov_b32 %s, %s" % (vr(MR[q2]), vr(MN[q2])))
    e("s_nop 0")
    for q2 in range(2):
        for u in range(2):
            e("v_mul_f32 %s, %s, %s" % (vr(L[q2][u]), vr(L[q2][u]), vr(AL[q2])))
    nt_ = 16
    for q2 in range(2):
        for base in range(0, 64, nt_):
            for k in range(nt_):
                e("v_accvgpr_read_b32 %s, a%d" % (vr(TMP[4 + k]), q2 * 64 + base + k))
            for k in range(nt_):
                e("v_mul_f32 %s, %s, %s" % (vr(TMP[4 + k]), vr(TMP[4 + k]), vr(AL[q2])))
            for k in range(nt_):
                e("v_accvgpr_write_b32 a%d, %s" % (q2 * 64 + base + k, vr(TMP[4 + k])))
    e("s_setpc_b64 %s" % sp(S_RET))
    # ---- store ----
    e("F4_store_%=:")
    for q2 in range(2):
        ta, tb = TMP[0], TMP[1]
        e("v_add_f32 %s, %s, %s" % (vr(ta), vr(L[q2][0]), vr(L[q2][1])))
        e("v_mov_b32 %s, %s" % (vr(tb), vr(ta)))
        e("s_nop 1")
        e("v_permlane32_swap_b32 %s, %s" % (vr(tb), vr(ta)))
        e("v_add_f32 %s, %s, %s" % (vr(ta), vr(ta), vr(tb)))
        e("v_rcp_f32 %s, %s" % (vr(AL[q2]), vr(ta)))
    e("s_nop 0")
    for q2 in range(2):
        e("s_mov_b64 exec, %s" % OP["rowmask%d" % q2])
        for db in range(4):
            # 16 accumulator registers -> 4 stores of 8 bytes
            for k in range(16):
                e("v_accvgpr_read_b32 %s, a%d" % (vr(TMP[4 + k]), O(q2, db) + k))
            for k in range(16):
                e("v_mul_f32 %s, %s, %s" % (vr(TMP[4 + k]), vr(TMP[4 + k]), vr(AL[q2])))
            for k in range(0, 16, 2):
                e("v_cvt_pk_bf16_f32 %s, %s, %s" % (vr(TMP[4 + k // 2]), vr(TMP[4 + k]), vr(TMP[4 + k + 1])))
            for g4 in range(4):
                e("global_store_dwordx2 %s, %s, %s offset:%d" % (OP["ooff%d" % q2], vr(TMP[4 + 2 * g4], 2), OP["obase"], (db * 32 + 8 * g4) * 2))
        e("s_mov_b64 exec, %s" % sp(S_EXEC))
    e("s_mov_b32 m0, s%d" % S_M0)
    return g_.lines


def main():
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ominix-mlx_amd", "csrc")
    if "--out" in sys.argv:      # (tests/test_generated_sources.py regenerates into a scratch directory and compares with the tree)
        out_dir = sys.argv[sys.argv.index("--out") + 1]
    variants = [(THR, (), "attn_flash4_body.inc"), (0.0, (), "attn_flash4_body_thr0.inc")]
    if "--diag" in sys.argv:     # timing-only builds for OMX_ATTN_W4_VAR=2..5 (attn_flash4.hip compiles them under -DOMX_F4_DIAG)
        variants += [(THR, ("nodma",), "attn_flash4_body_d2.inc"), (THR, ("novalu",), "attn_flash4_body_d3.inc"),
                     (THR, ("nolds",), "attn_flash4_body_d4.inc"), (THR, ("nodma", "novalu", "nolds"), "attn_flash4_body_d5.inc"),
                     (THR, ("expmov",), "attn_flash4_body_d6.inc"), (THR, ("nobar",), "attn_flash4_body_d7.inc"), (THR, ("noadd",), "attn_flash4_body_d8.inc")]
    for thr, diag, name in variants:
        lines = generate(thr, diag)
        with open(os.path.join(out_dir, name), "w") as f:
            f.write("// GENERATED by tools/gen_flash4_asm.py (deferred-rescale threshold %.1f) -- do not edit; %d instructions\n" % (thr, len(lines)))
            for ln in lines:
                f.write('"%s\\n\\t"\n' % ln)
    clob = ["v%d" % i for i in range(32, 256)] + ["a%d" % i for i in range(256)] + ["s%d" % i for i in range(60, 96)] + ["vcc", "scc", "memory"]
    with open(os.path.join(out_dir, "attn_flash4_clobbers.inc"), "w") as f:
        f.write("// GENERATED by tools/gen_flash4_asm.py -- registers the body owns\n")
        f.write("#define F4_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + "\n")
    print("operands:", ", ".join("%s=%s" % (n, OP[n]) for n, _ in OPERANDS))


if __name__ == "__main__":
    main()
