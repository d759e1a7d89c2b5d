"""Paraformer pass under rocprofv3 --kernel-trace: per pass, kernel count, summed kernel time and first-start-to-last-end span of the
encoder (mel_power .. cif_fire) and the decoder (after cif_fire .. argmax).  python tools/ktrace_stages.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    s = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    return s.split("(")[0].split("::")[-1]
k = [(nm(r), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
mels = [i for i, n in enumerate(k) if "mel_power" in n[0]]
cifs = [i for i, n in enumerate(k) if "cif_fire" in n[0]]
for p, (a, c) in enumerate(zip(mels, cifs)):
    end = mels[p + 1] if p + 1 < len(mels) else len(k)
    enc, dec = k[a:c], k[c + 1:end]
    last = [i for i, n in enumerate(dec) if "argmax" in n[0]]
    dec = dec[:last[0] + 1] if last else dec
    f = lambda seg: (len(seg), round(sum(e - s for _, s, e in seg) / 1e3, 1), round((seg[-1][2] - seg[0][1]) / 1e3, 1))
    print("pass", p, "encoder (kernels, busy us, span us)", f(enc), "decoder", f(dec), "gap before decoder us", round((dec[0][1] - k[c][2]) / 1e3, 1))
