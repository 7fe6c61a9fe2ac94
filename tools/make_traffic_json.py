"""Turn a tools/pmc_passes.sh summary.json of `encode_hidden` into profiles/r02_traffic.json (what bench.py reports as
roofline.traffic when the kernel source hash and the launch shape match).  Usage:
    python tools/make_traffic_json.py <pmc summary.json> <rows per launch> <out json> [encode_key]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summ, rows, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
c = {k: v["mean_per_dispatch"] for k, v in json.load(open(summ))["counters"].items()}
fused = len(sys.argv) > 4 and sys.argv[4] == "encode_key"
with open(os.path.join(ROOT, "coponerf_amd", "csrc", "encode_fused.hip" if fused else "encode.hip"), "rb") as f:
    sha = hashlib.sha256(f.read()).hexdigest()[:16]
fetch_kib, write_kib = c["FETCH_SIZE"], c["WRITE_SIZE"]
rec = {
    "_comment": "HBM traffic per launch of the first-layer kernel (encode_hidden_kernel, or encode_key_kernel with the folded key layer fused) from rocprofv3 PMC passes (tools/pmc_passes.sh), corrected as "
                "MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KiB) x 2 for wide coalesced reads, WRITE_SIZE (KiB) x 1.",
    "kernel": "encode_fused_kernel" if fused else "encode_hidden_kernel", "kernel_source_sha16": sha, "shape": {"M": rows, "N": 832},
    "fetch_size_kib": fetch_kib, "write_size_kib": write_kib,
    "hbm_read_bytes": fetch_kib * 1024 * 2, "hbm_write_bytes": write_kib * 1024,
    "hbm_bytes": fetch_kib * 1024 * 2 + write_kib * 1024,
    "algorithmic_bytes": dict({"hid": rows * 832 * 2}, **({"kh": rows // 2 * 256} if fused else {})),
    "counters": {k: c[k] for k in sorted(c)},
}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps({k: rec[k] for k in ("kernel_source_sha16", "hbm_read_bytes", "hbm_write_bytes", "hbm_bytes")}))
