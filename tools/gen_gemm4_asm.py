"""Generator of ominix-mlx_amd/csrc/gemm4_body.inc: the K loop of the asm-scheduled 256 x 256 bf16 GEMM tile (gemm.hip gemm_bf16_nt_asm_kernel) as
ONE inline-asm statement.  EIGHT waves, two per SIMD, 128 x 64 of the tile per wave = 8 accumulators of v_mfma_f32_32x32x16_bf16 (128 AGPRs,
compiler-allocated "+a" operands %0..%7, so the C++ epilogue reads them like any other value); the waves run FREE between one s_barrier
per half-step of 32 k.

The point of this kernel (EXPERIMENTS.md R5-3): a 256^2 tile needs 64 KiB of operands per 64 k, and what the LDS-DMA path
(global_load_lds_dwordx4) lands per CU stops near 37 GB/s -- one 1-KiB piece per ~53 cycles, whoever issues it: 64 pieces = 3 400 cycles
per K step against 2 048 cycles of MFMAs, the ceiling BOTH the hipcc-scheduled eight-phase kernel (1.29 PF) and an all-DMA form of this one
(1.20 PF) sit on.  So only X takes that path; W goes HBM -> VGPR (global_load_dwordx4, full 128-B lines) -> ds_write_b128 into the same
swizzled image: two load paths side by side.

LDS (128 KiB of tiles): [X buffer 0 | X buffer 1 | W buffer 0 | W buffer 1], each 256 rows x 64 k bf16 = rows of 128 B (a piece = 8 whole rows),
16-B chunk index ^= (row >> 1) & 7 on the SOURCE side and again on the read: conflict-free for ds_read_b128 of 32 rows x 2 chunks.
K step t (64 k) lives in buffer t & 1, half-step h = 2 t + s uses its chunks 4 s .. 4 s + 3 (k steps ks = 0, 1 of 16):
    start of h:   s_waitcnt lgkmcnt(0) (every fragment read and staged write of this wave retired)   [odd h: s_waitcnt vmcnt(0)]   s_barrier
    during h:     16 MFMAs | the fragments of h + 1 in gaps 0..7 (ks = 0 in place, each register refilled after its last MFMA; ks = 1 into the
                  alternate set) | odd h = 2 t + 1, gaps 8..15: this wave's 4 W loads and 4 X DMA pieces of K step t + 2 (buffer t & 1: its
                  last fragments were read during 2 t) | even h, gaps 8..11: s_waitcnt vmcnt(4) + the 4 staged W pieces -> LDS
K % 128 == 0 (pairs of K steps with static buffers); the last pair loads nothing, the last half-step reads no fragments.

Operands: %0..%7 accumulators acc[i * 2 + j] (row block i, column block j of the wave's 128 x 64), %8 LDS address of this thread's 16 parameter
dwords (source offsets of its 4 X and 4 W pieces, fragment read addresses per 32-k quarter of the row), %9 / %10 global bases of X / this
wave's W rows (64-bit, advanced here by 128 B per K step), %11 pairs of K steps, %12 LDS address of the tiles + wave * 4 KiB.
Fixed registers (clobbered): v[16:31] parameters, v32 staged-write address, v[36:51] W staging, v[52:75] ks = 0 fragments, v[76:99] / v[100:123]
ks = 1 fragment sets, s[60:65].
"""
import os
import sys

RI, CJ = 4, 2
ACC = lambda i, j: "%%%d" % (i * CJ + j)
P_ADDR, XBASE, WBASE, NPAIRS, LDSW = "%8", "%9", "%10", "%11", "%12"
PRM = 16
XDMA = [PRM + k for k in range(4)]
WDMA = [PRM + 4 + k for k in range(4)]
XFR = [PRM + 8 + k for k in range(4)]      # by quarter q = 2 s + ks of the 128-B row
WFR = [PRM + 12 + k for k in range(4)]
WLDS = 32
STAGE = 36
A0 = 52
B = [76, 100]
S_X, S_W, S_LOOP, S_M0 = 60, 62, 64, 65


def vr(lo, n=1):
    return "v%d" % lo if n == 1 else "v[%d:%d]" % (lo, lo + n - 1)
def sp(lo):
    return "s[%d:%d]" % (lo, lo + 1)
def frag_reg(bset, mat, idx, ks):
    base = A0 if ks == 0 else B[bset]
    return base + (idx * 4 if mat == 0 else 16 + idx * 4)


class Gen(list):
    def __init__(self, diag=()):
        super().__init__()
        self.diag = set(diag)
        self.lds = []
        self.nid = 0
    def e(self, s):
        op = s.split()[0]
        if "nodma" in self.diag and (op in ("global_load_lds_dwordx4", "global_load_dwordx4") or s.startswith("s_add_u32 m0")):
            return
        if "nolds" in self.diag and op in ("ds_read_b128", "ds_write_b128") and "PARAM" not in s:
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


def dma_x(g, buf, it):
    g.e("s_add_u32 m0, %s, %d" % (LDSW, buf * 32768 + it * 1024))
    return "global_load_lds_dwordx4 %s, %s" % (vr(XDMA[it]), sp(S_X))

def load_w(g, it):
    g.e("global_load_dwordx4 %s, %s, %s" % (vr(STAGE + 4 * it, 4), vr(WDMA[it]), sp(S_W)))

def write_w(g, buf, it):
    return g.lds_op("ds_write_b128 %s, %s offset:%d" % (vr(WLDS), vr(STAGE + 4 * it, 4), buf * 32768 + it * 1024))

def advance(g, s):
    g.e("s_add_u32 s%d, s%d, 128" % (s, s))
    g.e("s_addc_u32 s%d, s%d, 0" % (s + 1, s + 1))

def read_frag(g, h, mat, idx, ks, bset):
    """fragment (block idx, k step ks) of half-step h (position in a pair of K steps: 0..3); ks = 1 goes to set bset"""
    buf, s = (h >> 1) & 1, h & 1
    reg = (WFR if mat else XFR)[2 * s + ks]
    return g.lds_op("ds_read_b128 %s, %s offset:%d" % (vr(frag_reg(bset, mat, idx, ks), 4), vr(reg), buf * 32768 + idx * 4096))


def half_step(g, p, loads, reads, writes, fr):
    """half-step p (0..3 inside a pair of K steps: buffer p >> 1, half p & 1).  fr: {(mat, idx, ks): LDS-op id} of this half-step's fragments
    (read during the previous one); returns the same for the next half-step."""
    g.lds_wait_all()
    if p & 1:
        g.e("s_waitcnt vmcnt(0)")
    g.e("s_barrier")
    bcur, bnxt = p & 1, (p + 1) & 1
    nxt = {}
    # reads of half-step p + 1 by gap: ks = 1 fragments (alternate set) early, ks = 0 in place after the register's last MFMA
    plan = {k: [] for k in range(16)}
    if reads:
        early = [(0, i, 1) for i in range(RI)] + [(1, j, 1) for j in range(CJ)]
        for k, f in enumerate(early):
            plan[k].append(f)
        for i in range(RI):
            plan[2 * i + 1].append((0, i, 0))       # X[i][0]: MFMAs 2 i, 2 i + 1
        plan[6].append((1, 0, 0))                   # W[0][0]: last MFMA 6
        plan[7].append((1, 1, 0))                   # W[1][0]: last MFMA 7
    for gap in range(16):
        ks, i, j = gap >> 3, (gap >> 1) & 3, gap & 1
        g.lds_wait([fr[(0, i, ks)], fr[(1, j, ks)]])
        g.e("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (ACC(i, j), vr(frag_reg(bcur, 1, j, ks), 4), vr(frag_reg(bcur, 0, i, ks), 4), ACC(i, j)))
        for (mat, idx, k2) in plan[gap]:
            nxt[(mat, idx, k2)] = read_frag(g, (p + 1) & 3, mat, idx, k2, bnxt)
        if loads and 8 <= gap < 12:
            load_w(g, gap - 8)
            if gap == 11:
                advance(g, S_W)
        if loads and gap >= 12:
            ld = dma_x(g, p >> 1, gap - 12)
            g.e("s_nop 0")
            g.e(ld)
            if gap == 15:
                advance(g, S_X)
        if writes and 8 <= gap < 12:
            if gap == 8:
                g.e("s_waitcnt vmcnt(4)")
            write_w(g, ((p >> 1) + 1) & 1, gap - 8)
    return nxt


def generate(diag=()):
    g = Gen(diag)
    g.e("s_mov_b32 s%d, m0" % S_M0)
    for k in range(4):
        g.e("ds_read_b128 %s, %s offset:%d ;PARAM" % (vr(PRM + 4 * k, 4), P_ADDR, 16 * k))
    g.e("s_mov_b64 %s, %s" % (sp(S_X), XBASE))
    g.e("s_mov_b64 %s, %s" % (sp(S_W), WBASE))
    # staged-write address: W region + this wave's 4 KiB + lane * 16
    g.e("v_mbcnt_lo_u32_b32 %s, -1, 0" % vr(WLDS))
    g.e("v_mbcnt_hi_u32_b32 %s, -1, %s" % (vr(WLDS), vr(WLDS)))
    g.e("v_lshlrev_b32 %s, 4, %s" % (vr(WLDS), vr(WLDS)))
    g.e("v_add_u32 %s, %s, %s" % (vr(WLDS), LDSW, vr(WLDS)))
    g.e("v_add_u32 %s, 0x10000, %s" % (vr(WLDS), vr(WLDS)))
    g.e("s_waitcnt lgkmcnt(0)")
    for buf in range(2):             # K steps 0 and 1
        for it in range(4):
            load_w(g, it)
        advance(g, S_W)
        for it in range(4):
            ld = dma_x(g, buf, it)
            g.e("s_nop 0")
            g.e(ld)
        advance(g, S_X)
        g.e("s_waitcnt vmcnt(4)")
        for it in range(4):
            write_w(g, buf, it)
    g.e("s_waitcnt vmcnt(0)")
    g.lds_wait_all()
    g.e("s_barrier")
    fr = {}
    for ks in range(2):
        for idx in range(RI):
            fr[(0, idx, ks)] = read_frag(g, 0, 0, idx, ks, 0)
        for idx in range(CJ):
            fr[(1, idx, ks)] = read_frag(g, 0, 1, idx, ks, 0)
    g.e("s_sub_u32 s%d, %s, 1" % (S_LOOP, NPAIRS))
    g.e("s_cmp_eq_u32 s%d, 0" % S_LOOP)
    g.e("s_cbranch_scc1 G8_tail_%=")
    g.lds_wait_all()                 # (the loop's first half-step starts from a known state whichever way it is entered)
    g.e("G8_loop_%=:")
    fr_loop = dict(fr)
    for p in range(4):
        fr_loop = half_step(g, p, p & 1, True, not (p & 1), fr_loop)
    g.lds_wait_all()
    g.e("s_sub_u32 s%d, s%d, 1" % (S_LOOP, S_LOOP))
    g.e("s_cmp_lg_u32 s%d, 0" % S_LOOP)
    g.e("s_cbranch_scc1 G8_loop_%=")
    g.e("G8_tail_%=:")
    g.lds_wait_all()
    fr_t = dict(fr)
    for p in range(4):
        fr_t = half_step(g, p, False, p < 3, p == 0, fr_t)
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
    with open(os.path.join(out_dir, "gemm4_body.inc"), "w") as f:
        lines = generate()
        f.write("// GENERATED by tools/gen_gemm4_asm.py -- do not edit; %d instructions\n" % len(lines))
        clob = ["v%d" % i for i in range(PRM, 124)] + ["s%d" % i for i in range(60, 66)] + ["scc", "memory"]
        f.write("#define G8_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + "\n")
        emit(f, "G8_BODY", lines)
        if "--diag" in sys.argv:
            for k, d in enumerate((("nodma",), ("nolds",), ("nobar",), ("nodma", "nolds", "nobar")), 1):
                emit(f, "G8_BODY_D%d" % k, generate(d))
        print("%d instructions" % len(lines))


if __name__ == "__main__":
    main()
