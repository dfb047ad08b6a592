"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round3.sh (or profile_round2.sh) into profiles/r0N_traffic.json, stamped with the sha256 of the
library they were measured on (bench.py reports `roofline.traffic` only when that stamp matches the library it runs).

    python tools/make_traffic_json.py gpurun_out/prof_r03 profiles/r03_traffic.json
"""
import hashlib
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def means(path, counter):
    out = []
    for line in open(path):
        m = re.match(rf"{counter}\s+n=\s*\d+\s+mean=\s*([0-9.]+)", line)
        if m:
            out.append(float(m.group(1)))
    return out


def entry(fetch_kib, write_kib, algorithmic):
    return {"FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
            "traffic_bytes": int(round((2.0 * fetch_kib + write_kib) * 1024)), "algorithmic_bytes": int(algorithmic)}


def gemm_bytes(m, n, k, x_bytes=2, y_bytes=2):
    return x_bytes * m * k + n * (k // 4) + 8 * n + y_bytes * m * n


g = os.path.join(src, "gemm_4096_pmc_summary.txt")
r = os.path.join(src, "rows_decode_pmc_summary.txt")
f, w = means(r, "FETCH_SIZE"), means(r, "WRITE_SIZE")
# order of the passes in profile_round2.sh step 4: staged 16x4096^2, shared 16x28672x4096, shared 16x16384^2, gemv 1x16384^2, pack 4096^2
names = [("gemm_rows_staged_16x4096x4096_f16", gemm_bytes(16, 4096, 4096)), ("gemm_rows_shared_16x28672x4096_f16", gemm_bytes(16, 28672, 4096)),
         ("gemm_rows_shared_16x16384x16384_f16", gemm_bytes(16, 16384, 16384)), ("gemv_1x16384x16384_f16", gemm_bytes(1, 16384, 16384)),
         ("quantize_pack_4096x4096_f32", 4 * 4096 * 4096 + 4096 * 1024)]
assert len(f) == len(names) and len(w) == len(names), (f, w)
with open(os.path.join(root, "ternary-vllm_amd", "libternary_hip.so"), "rb") as fh:
    sha = hashlib.sha256(fh.read()).hexdigest()
doc = {"note": "HBM-side bytes per launch from rocprofv3 --pmc passes (one pass per counter; tools/profile_round3.sh): traffic = 2 x FETCH_SIZE "
               "(gfx950 correction: the counter reports 64-byte units as 32) + WRITE_SIZE, both in KiB per dispatch, mean over the dispatches of "
               "the pass.  The split-K kernels' entries cover the main kernel only (the slab reduce is a second dispatch).",
       "library_sha256": sha,
       "gemm_4096x4096x4096_f16": entry(means(g, "FETCH_SIZE")[0], means(g, "WRITE_SIZE")[0], gemm_bytes(4096, 4096, 4096))}
for (name, alg), fk, wk in zip(names, f, w):
    doc[name] = entry(fk, wk, alg)
with open(dst, "w") as fh:
    json.dump(doc, fh, indent=1)
print(json.dumps(doc, indent=1))
