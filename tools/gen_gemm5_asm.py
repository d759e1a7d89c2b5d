"""Generator of ominix-mlx_amd/csrc/gemm5_body.inc: the K loop of the FOUR-wave 256 x 256 bf16 / f16 GEMM tile (gemm.hip gemm_nt_w4_kernel) as
ONE inline-asm statement.

Geometry (EXPERIMENTS.md R5-4): one wave per SIMD owns the whole 512-register file and 128 x 128 of the tile = 8 x 8 accumulators of
v_mfma_f32_16x16x32 (256 AGPRs: a[(i * 8 + j) * 4 ..+3] is row block i, column block j; bound to the C++ side as sixteen physical-register
operands "+{a[16 o : 16 o + 15]}", so the epilogue reads them like any value), plus 128 VGPRs holding the fragments of a WHOLE 64-k step
(two halves of 32 k).  A wave of 128 x 128 reads (128 + 128) rows x 64 k from LDS per step, 128 KiB per CU for the four of them; eight waves
of 128 x 64 read 192 KiB.  The 16x16x32 instruction, not the 32x32x16 one: the same kernel runs 15-18 % slower on the latter on a full chip
(power, not issue: both reach the same rate on a few CUs).

LDS (128 KiB of tiles): [X buffer 0 | X 1 | W 0 | W 1], each 256 rows x 64 k = rows of 128 B; a DMA piece = 8 whole rows = 1 KiB (lane -> row
lane >> 3, 16-B slot lane & 7), the 16-B chunk index XORed with (row >> 1) & 7 on the SOURCE side and again on the read: ds_read_b128 of 16 rows
x 4 chunks is conflict-free (tools/probes/lds_b128_probe.hip).  Wave w stages pieces 8 w .. 8 w + 7 of either operand (rows 64 w .. 64 w + 63).

K step t (buffer t & 1), 128 MFMAs in gaps 0..127 (k half kh = gap >> 6 of 32 k; row block i = (gap >> 3) & 7, column block j = gap & 7):
    gaps 0..14    (every other) read the X fragments of the second half -- the first half's are in registers since the previous step
    after gap 20  lgkmcnt(0), s_barrier: every wave is done with X buffer t & 1  ->  gaps 22..37: this wave's 8 X pieces of step t + 2 by DMA into
                  it, alternating with the W fragments of the second half
    after gap 42  lgkmcnt(0), s_barrier: likewise W  ->  gaps 44..58 (every other): the 8 W pieces of step t + 2
    gap LAND      vmcnt(16) (this wave's pieces of step t + 1 have landed; the 16 of t + 2 may be in flight), s_barrier, then every other gap reads
                  one of the 16 first-half fragments of step t + 1 from the other buffer (their registers were last used by gap 63)
Two steps per loop trip (static buffer offsets), K % 128 == 0, at least two steps; the last two steps load nothing.

Operands: %0..%15 the accumulators (named by their physical registers here); %16 LDS address of this thread's 20 parameter dwords (DMA
source offsets of its 8 X and 8 W pieces, fragment read addresses per k half); %17 / %18 global bases of X / of this wave's W rows (advanced
by 128 B per step); %19 loop trips ((K / 64 - 2) / 2); %20 LDS address of the tiles + wave * 8 KiB.
Fixed registers (clobbered): v[16:35] parameters, v[40:103] first-half fragments (X then W), v[104:167] second-half, s[60:65]."""
import os
import sys

NJ = 8
PRM = 16
H = [40, 104]                               # fragment halves: X [NI i] x 4 regs, then (from + 32) W [8 j]
S_X, S_W, S_LOOP, S_M0 = 60, 62, 64, 65
LAST_VGPR = 167


class Geo:
    """tile rows 256: 8 row blocks of 16 per wave, 8 X pieces per wave; 128 (the half-height tile for grids of exactly one tile per CU):
    4 row blocks, 4 X pieces, X buffers of 16 KiB.  Operand numbers follow the accumulator count (16 or 8 operands of 16 AGPRs)."""
    def __init__(self, rows):
        self.rows = rows
        self.NI = rows // 32
        self.XP = rows // 32                 # X pieces per wave
        self.XB = rows * 128                 # bytes of an X buffer
        self.WOFF = 2 * self.XB              # W buffers follow the two X buffers
        self.G = self.NI * NJ * 2            # MFMAs per K step and wave
        self.st = 2 if rows == 256 else 1    # spacing of the sparse filler runs
        n = self.NI * NJ * 4 // 16           # accumulator operands
        self.P_ADDR, self.XBASE, self.WBASE, self.NLOOP, self.LDSX, self.LDSWW, self.WSEL = ["%%%d" % (n + k) for k in range(7)]
        self.XDMA = [PRM + k for k in range(self.XP)]
        self.WDMA = [PRM + self.XP + k for k in range(8)]
        self.XFR = [PRM + self.XP + 8 + k for k in range(2)]      # by k half
        self.WFR = [PRM + self.XP + 10 + k for k in range(2)]
        self.NPRM = self.XP + 12
TWO_SCHEDULES = os.environ.get("G5_TWO", "0") == "1"     # odd waves on a schedule shifted by one gap (operand: wave & 1)
DENSE = os.environ.get("G5_DENSE", "0") == "1"
ONE_BARRIER = os.environ.get("G5_ONE_BARRIER", "0") == "1"   # one barrier per step for the refill of both operands instead of two
W_OUTER = os.environ.get("G5_ORDER", "ij") == "ji"   # the MFMA order inside a k half: column block outer (srcA = the W fragment stays), row block inner
LAND = int(os.environ["G5_LAND"]) if "G5_LAND" in os.environ else None     # the gap of the landing wait (default: two gaps into the second k half)


def vr(lo, n=1):
    return "v%d" % lo if n == 1 else "v[%d:%d]" % (lo, lo + n - 1)
def sp(lo):
    return "s[%d:%d]" % (lo, lo + 1)
def acc(i, j):
    r = (i * NJ + j) * 4
    return "a[%d:%d]" % (r, r + 3)
def frag(mat, kh, idx):
    return H[kh] + mat * 32 + idx * 4


class Gen(list):
    def __init__(self, mfma, geo, diag=()):
        super().__init__()
        self.mfma = mfma
        self.geo = geo
        self.diag = set(diag)
        self.lds = []
        self.nid = 0
    def e(self, s):
        op = s.split()[0]
        if "nodma" in self.diag and (op == "global_load_lds_dwordx4" or s.startswith("s_add_u32 m0")):
            return
        if "nolds" in self.diag and op == "ds_read_b128" and "PARAM" not in s:
            return
        if "nobar" in self.diag and (op == "s_barrier" or s.startswith("s_waitcnt vmcnt")):
            return
        self.append(s.replace(" ;PARAM", ""))
    def lds_op(self, s):
        self.e(s)
        self.lds.append(self.nid)
        self.nid += 1
        return self.nid - 1
    def lds_wait(self, ids):
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


def dma(g, mat, buf, it):
    q = g.geo
    if mat:
        g.e("s_add_u32 m0, %s, %d" % (q.LDSWW, buf * 32768 + it * 1024))
    else:
        g.e("s_add_u32 m0, %s, %d" % (q.LDSX, buf * q.XB + it * 1024))
    g.e("s_nop 0")
    g.e("global_load_lds_dwordx4 %s, %s" % (vr((q.WDMA if mat else q.XDMA)[it]), sp(S_W if mat else S_X)))

def advance(g, s):
    g.e("s_add_u32 s%d, s%d, 128" % (s, s))
    g.e("s_addc_u32 s%d, s%d, 0" % (s + 1, s + 1))

def read_frag(g, buf, mat, kh, idx):
    q = g.geo
    return g.lds_op("ds_read_b128 %s, %s offset:%d" % (vr(frag(mat, kh, idx), 4), vr((q.WFR if mat else q.XFR)[kh]),
                                                       buf * (32768 if mat else q.XB) + idx * 2048))


def step(g, buf, fr, loads, next_reads, vm, shift=0):
    """one K step out of buffer `buf`.  fr: {(mat, kh, idx): LDS-op id} of the first-half fragments already requested; returns the same for the
    next step.  loads: the DMA of step t + 2; next_reads: the first-half fragments of t + 1; vm: the vmcnt of the landing wait."""
    q = g.geo
    NI, G, st = q.NI, q.G, q.st
    fr = dict(fr)
    second_x = [(0, 1, i) for i in range(NI)]
    second_w = [(1, 1, j) for j in range(NJ)]
    if W_OUTER:
        first = [(1, 0, 0)] + [(0, 0, i) for i in range(NI)] + [(1, 0, j) for j in range(1, NJ)]
    else:
        first = [(0, 0, 0)] + [(1, 0, j) for j in range(NJ)] + [(0, 0, i) for i in range(1, NI)]
    # filler plan: gap -> list of actions
    plan = {}
    def at(gap, act):
        plan.setdefault(gap + shift, []).append(act)      # (shift: the odd waves' schedule, one gap behind the even waves')
    xs = 1 if DENSE else st              # spacing of the first run of reads (dense: the X buffer is released, and refilled, ~10 gaps earlier)
    for k in range(NI):
        at(k * xs, ("read", second_x[k]))
    if loads and ONE_BARRIER:
        # one barrier for both operands: all second-half fragments first (X on even, W on odd gaps), then the 16 pieces
        for k in range(NJ):
            at(k * st + (1 if st == 2 else NI), ("read", second_w[k]))
        last = max((NI - 1) * st, (NJ - 1) * st + (1 if st == 2 else NI))
        b1 = last + 2 * st + 2
        at(b1, ("bar",))
        pieces = [("dma", 0, k) for k in range(q.XP)] + [("dma", 1, k) for k in range(8)]
        for k, act in enumerate(pieces):
            at(b1 + st + k * st, act)
        assert b1 + st + (len(pieces) - 1) * st + shift < G // 2 + 2 + shift
    elif loads:
        b1 = (NI - 1) * xs + 3 * st
        at(b1, ("bar",))
        mixed = []
        for k in range(max(q.XP, NJ)):
            if k < q.XP:
                mixed.append(("dma", 0, k))
            if k < NJ:
                mixed.append(("read", second_w[k]))
        for k, act in enumerate(mixed):
            at(b1 + st + k, act)
        last = b1 + st + len(mixed) - 1
        b2 = last + 2 * st + 1
        at(b2, ("bar",))
        for k in range(8):
            at(b2 + st + k * st, ("dma", 1, k))
        assert b2 + st + 7 * st + shift < G // 2 + 2 + shift, "the step's DMA must be issued before the landing wait (vmcnt counts on it)"
    else:
        for k in range(NJ):
            at(NI * xs + k * st, ("read", second_w[k]))
    land = G // 2 + 2 if LAND is None else LAND
    if next_reads:
        at(land, ("land", vm))
        for k, f in enumerate(first):
            at(land + k * st, ("readnext", f))
        assert land + (len(first) - 1) * st + shift < G
    nxt = {}
    xdma = wdma = 0
    for gap in range(G):
        kh, i, j = gap // (G // 2), (gap % (G // 2)) // NJ, gap % NJ
        if W_OUTER:
            i, j = (gap % (G // 2)) % NI, (gap % (G // 2)) // NI
        g.lds_wait([fr[(0, kh, i)], fr[(1, kh, j)]])
        g.e("%s %s, %s, %s, %s" % (g.mfma, acc(i, j), vr(frag(1, kh, j), 4), vr(frag(0, kh, i), 4), acc(i, j)))
        for act in plan.get(gap, []):
            if act[0] == "read":
                fr[act[1]] = read_frag(g, buf, *act[1])
            elif act[0] == "readnext":
                nxt[act[1]] = read_frag(g, buf ^ 1, *act[1])
            elif act[0] == "bar":
                g.lds_wait_all()
                g.e("s_barrier")
            elif act[0] == "land":
                g.e("s_waitcnt vmcnt(%d)" % act[1])
                g.e("s_barrier")
            elif act[0] == "dma":
                dma(g, act[1], buf, act[2])
                if act[1] == 0:
                    xdma += 1
                    if xdma == q.XP:
                        advance(g, S_X)
                else:
                    wdma += 1
                    if wdma == 8:
                        advance(g, S_W)
    g.lds_wait_all()
    return nxt


def generate(mfma, rows=256, diag=()):
    q = Geo(rows)
    g = Gen(mfma, q, diag)
    g.e("s_mov_b32 s%d, m0" % S_M0)
    for k in range(q.NPRM // 4):
        g.e("ds_read_b128 %s, %s offset:%d ;PARAM" % (vr(PRM + 4 * k, 4), q.P_ADDR, 16 * k))
    g.e("s_mov_b64 %s, %s" % (sp(S_X), q.XBASE))
    g.e("s_mov_b64 %s, %s" % (sp(S_W), q.WBASE))
    g.e("s_mov_b32 s%d, %s" % (S_LOOP, q.NLOOP))
    g.e("s_waitcnt lgkmcnt(0)")
    for buf in range(2):             # K steps 0 and 1
        for mat in range(2):
            for it in range(8 if mat else q.XP):
                dma(g, mat, buf, it)
            advance(g, S_W if mat else S_X)
    per_step = q.XP + 8
    g.e("s_waitcnt vmcnt(%d)" % per_step)
    g.e("s_barrier")
    fr = {}
    for idx in range(q.NI):
        fr[(0, 0, idx)] = read_frag(g, 0, 0, 0, idx)
    for idx in range(NJ):
        fr[(1, 0, idx)] = read_frag(g, 0, 1, 0, idx)
    g.lds_wait_all()
    ready = {k: -1 for k in fr}      # (every step ends with lgkmcnt(0): the first-half fragments are in registers at its start)
    def loops(tag, shift):
        g.e("s_cmp_eq_u32 s%d, 0" % S_LOOP)
        g.e("s_cbranch_scc1 G5_tail%s_%%=" % tag)
        g.e("G5_loop%s_%%=:" % tag)
        step(g, 0, ready, True, True, per_step, shift)
        step(g, 1, ready, True, True, per_step, shift)
        g.e("s_sub_u32 s%d, s%d, 1" % (S_LOOP, S_LOOP))
        g.e("s_cmp_lg_u32 s%d, 0" % S_LOOP)
        g.e("s_cbranch_scc1 G5_loop%s_%%=" % tag)
        g.e("G5_tail%s_%%=:" % tag)
        step(g, 0, ready, False, True, 0, shift)
        step(g, 1, ready, False, False, 0, shift)
    if TWO_SCHEDULES:
        # odd waves run the same step with every filler one gap later: the four waves of a workgroup execute in lockstep between barriers,
        # and with ONE schedule their LDS reads and DMA issues land in the same cycles
        g.e("s_cmp_eq_u32 %s, 0" % q.WSEL)
        g.e("s_cbranch_scc0 G5_odd_%=")
        loops("", 0)
        g.e("s_branch G5_done_%=")
        g.e("G5_odd_%=:")
        loops("B", 1)
        g.e("G5_done_%=:")
    else:
        loops("", 0)
    g += ["s_nop 7", "s_nop 7", "s_nop 7"]      # (hipcc reads the accumulators next and does not know they come from MFMAs)
    g.e("s_mov_b32 m0, s%d" % S_M0)
    return g


def emit(f, name, lines):
    f.write("#define %s \\\n" % name)
    for ln in lines:
        f.write('    "%s\\n\\t" \\\n' % ln)
    f.write('    ""\n')


def main():
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ominix-mlx_amd", "csrc")
    if "--out" in sys.argv:      # (tests/test_generated_sources.py regenerates into a scratch directory and compares with the tree)
        out_dir = sys.argv[sys.argv.index("--out") + 1]
    with open(os.path.join(out_dir, "gemm5_body.inc"), "w") as f:
        lines = generate("v_mfma_f32_16x16x32_bf16")
        half = generate("v_mfma_f32_16x16x32_bf16", 128)
        f.write("// GENERATED by tools/gen_gemm5_asm.py -- do not edit; %d instructions (256-row tile), %d (128-row tile)\n" % (len(lines), len(half)))
        clob = ["v%d" % i for i in range(PRM, LAST_VGPR + 1)] + ["s%d" % i for i in range(60, 66)] + ["scc", "memory"]
        f.write("#define G5_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + "\n")
        emit(f, "G5_BODY", lines)
        emit(f, "G5_BODY_F16", generate("v_mfma_f32_16x16x32_f16"))
        emit(f, "G5H_BODY", half)
        emit(f, "G5H_BODY_F16", generate("v_mfma_f32_16x16x32_f16", 128))
        if "--diag" in sys.argv:
            for k, d in enumerate((("nodma",), ("nolds",), ("nobar",), ("nodma", "nolds", "nobar"), ("nodma", "nobar"), ("nolds", "nobar"),
                                   ("nodma", "nolds")), 1):
                emit(f, "G5_BODY_D%d" % k, generate("v_mfma_f32_16x16x32_bf16", 256, d))
        print("%d / %d instructions" % (len(lines), len(half)))


if __name__ == "__main__":
    main()
