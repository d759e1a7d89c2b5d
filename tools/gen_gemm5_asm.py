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

NI = NJ = 8
P_ADDR, XBASE, WBASE, NLOOP, LDSW = "%16", "%17", "%18", "%19", "%20"
PRM = 16
XDMA = [PRM + k for k in range(8)]
WDMA = [PRM + 8 + k for k in range(8)]
XFR = [PRM + 16 + k for k in range(2)]      # by k half
WFR = [PRM + 18 + k for k in range(2)]
H = [40, 104]                               # fragment halves: X [8 i] x 4 regs, then W [8 j]
S_X, S_W, S_LOOP, S_M0 = 60, 62, 64, 65
LAST_VGPR = 167
W_OUTER = os.environ.get("G5_ORDER", "ij") == "ji"   # the MFMA order inside a k half: column block outer (srcA = the W fragment stays), row block inner
LAND = int(os.environ.get("G5_LAND", "66"))     # the gap of the landing wait (first-half registers are free from gap 64)


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
    def __init__(self, mfma, diag=()):
        super().__init__()
        self.mfma = mfma
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
    g.e("s_add_u32 m0, %s, %d" % (LDSW, mat * 65536 + buf * 32768 + it * 1024))
    g.e("s_nop 0")
    g.e("global_load_lds_dwordx4 %s, %s" % (vr((WDMA if mat else XDMA)[it]), sp(S_W if mat else S_X)))

def advance(g, s):
    g.e("s_add_u32 s%d, s%d, 128" % (s, s))
    g.e("s_addc_u32 s%d, s%d, 0" % (s + 1, s + 1))

def read_frag(g, buf, mat, kh, idx):
    return g.lds_op("ds_read_b128 %s, %s offset:%d" % (vr(frag(mat, kh, idx), 4), vr((WFR if mat else XFR)[kh]), buf * 32768 + idx * 2048))


def step(g, buf, fr, loads, next_reads, vm):
    """one K step out of buffer `buf`.  fr: {(mat, kh, idx): LDS-op id} of the first-half fragments already requested; returns the same for the
    next step.  loads: the DMA of step t + 2; next_reads: the first-half fragments of t + 1; vm: the vmcnt of the landing wait."""
    fr = dict(fr)
    second_x = [(0, 1, i) for i in range(NI)]
    second_w = [(1, 1, j) for j in range(NJ)]
    if W_OUTER:
        first = [(1, 0, 0)] + [(0, 0, i) for i in range(NI)] + [(1, 0, j) for j in range(1, NJ)]
    else:
        first = [(0, 0, 0)] + [(1, 0, j) for j in range(NJ)] + [(0, 0, i) for i in range(1, NI)]
    nxt = {}
    for gap in range(128):
        kh, i, j = gap >> 6, (gap >> 3) & 7, gap & 7
        if W_OUTER:
            i, j = j, i
        g.lds_wait([fr[(0, kh, i)], fr[(1, kh, j)]])
        g.e("%s %s, %s, %s, %s" % (g.mfma, acc(i, j), vr(frag(1, kh, j), 4), vr(frag(0, kh, i), 4), acc(i, j)))
        if gap < 16 and gap % 2 == 0:
            f = second_x[gap // 2]
            fr[f] = read_frag(g, buf, *f)
        if loads:
            if gap == 20 or gap == 42:
                g.lds_wait_all()
                g.e("s_barrier")
            if 22 <= gap < 38:
                if gap % 2 == 0:
                    dma(g, 0, buf, (gap - 22) // 2)
                else:
                    f = second_w[(gap - 23) // 2]
                    fr[f] = read_frag(g, buf, *f)
                if gap == 36:
                    advance(g, S_X)
            if 44 <= gap < 60 and gap % 2 == 0:
                dma(g, 1, buf, (gap - 44) // 2)
                if gap == 58:
                    advance(g, S_W)
        elif 16 <= gap < 32 and gap % 2 == 0:
            f = second_w[(gap - 16) // 2]
            fr[f] = read_frag(g, buf, *f)
        if next_reads:
            if gap == LAND:
                g.e("s_waitcnt vmcnt(%d)" % vm)
                g.e("s_barrier")
            if LAND <= gap < LAND + 32 and (gap - LAND) % 2 == 0:
                f = first[(gap - LAND) // 2]
                nxt[f] = read_frag(g, buf ^ 1, *f)
    g.lds_wait_all()
    return nxt


def generate(mfma, diag=()):
    g = Gen(mfma, diag)
    g.e("s_mov_b32 s%d, m0" % S_M0)
    for k in range(5):
        g.e("ds_read_b128 %s, %s offset:%d ;PARAM" % (vr(PRM + 4 * k, 4), P_ADDR, 16 * k))
    g.e("s_mov_b64 %s, %s" % (sp(S_X), XBASE))
    g.e("s_mov_b64 %s, %s" % (sp(S_W), WBASE))
    g.e("s_mov_b32 s%d, %s" % (S_LOOP, NLOOP))
    g.e("s_waitcnt lgkmcnt(0)")
    for buf in range(2):             # K steps 0 and 1
        for mat in range(2):
            for it in range(8):
                dma(g, mat, buf, it)
            advance(g, S_W if mat else S_X)
    g.e("s_waitcnt vmcnt(16)")
    g.e("s_barrier")
    fr = {}
    for idx in range(NI):
        fr[(0, 0, idx)] = read_frag(g, 0, 0, 0, idx)
    for idx in range(NJ):
        fr[(1, 0, idx)] = read_frag(g, 0, 1, 0, idx)
    g.lds_wait_all()
    ready = {k: -1 for k in fr}      # (every step ends with lgkmcnt(0): the first-half fragments are in registers at its start)
    g.e("s_cmp_eq_u32 s%d, 0" % S_LOOP)
    g.e("s_cbranch_scc1 G5_tail_%=")
    g.e("G5_loop_%=:")
    step(g, 0, ready, True, True, 16)
    step(g, 1, ready, True, True, 16)
    g.e("s_sub_u32 s%d, s%d, 1" % (S_LOOP, S_LOOP))
    g.e("s_cmp_lg_u32 s%d, 0" % S_LOOP)
    g.e("s_cbranch_scc1 G5_loop_%=")
    g.e("G5_tail_%=:")
    step(g, 0, ready, False, True, 0)
    step(g, 1, ready, False, False, 0)
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
    with open(os.path.join(out_dir, "gemm5_body.inc"), "w") as f:
        lines = generate("v_mfma_f32_16x16x32_bf16")
        f.write("// GENERATED by tools/gen_gemm5_asm.py -- do not edit; %d instructions\n" % len(lines))
        clob = ["v%d" % i for i in range(PRM, LAST_VGPR + 1)] + ["s%d" % i for i in range(60, 66)] + ["scc", "memory"]
        f.write("#define G5_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + "\n")
        emit(f, "G5_BODY", lines)
        emit(f, "G5_BODY_F16", generate("v_mfma_f32_16x16x32_f16"))
        if "--diag" in sys.argv:
            for k, d in enumerate((("nodma",), ("nolds",), ("nobar",), ("nodma", "nolds", "nobar"), ("nodma", "nobar"), ("nolds", "nobar"),
                                   ("nodma", "nolds")), 1):
                emit(f, "G5_BODY_D%d" % k, generate("v_mfma_f32_16x16x32_bf16", d))
        print("%d instructions" % len(lines))


if __name__ == "__main__":
    main()
