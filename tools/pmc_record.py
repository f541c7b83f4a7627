"""Turn the two PMC passes over tools/roofline_only.py into profiles/pmc_decode_attn_<kernel-source-hash>.json, the
file dimx.roofline.pmc_traffic reads for bench.py's `roofline.traffic`.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python tools/roofline_only.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python tools/roofline_only.py
    python tools/pmc_record.py gpurun_out/pmc_fetch gpurun_out/pmc_write [commit]

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (16-B/lane streaming reads on gfx950 report half of the
bytes, MI355X_MICROARCH.md section HBM)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dimx  # noqa: E402,F401
from dimx import roofline  # noqa: E402

KERNELS = (("decode_attn_kernel<dimx::bf16, false, true, 1>", "pmc_decode_attn_%s.json", roofline.DECODE_ATTN_SOURCES),
           ("xcd_layer_kernel", "pmc_layer_chain_%s.json", roofline.LAYER_CHAIN_SOURCES))


def avg_counter(d, counter, kernel):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for %s under %s" % (counter, kernel, d))
    return sum(vals) / len(vals), len(vals)


def main():
    for kernel, fname, sources in KERNELS:
        fetch_kb, n1 = avg_counter(sys.argv[1], "FETCH_SIZE", kernel)
        write_kb, n2 = avg_counter(sys.argv[2], "WRITE_SIZE", kernel)
        rec = {"kernel": kernel, "B": 256, "T": 300, "mode": "bf16", "commit": sys.argv[3] if len(sys.argv) > 3 else None,
               "kernel_source_sha256_12": roofline.kernel_source_hash(sources), "FETCH_SIZE_KB_avg": fetch_kb, "launches_fetch": n1,
               "WRITE_SIZE_KB_avg": write_kb, "launches_write": n2,
               "traffic_bytes": (2.0 * fetch_kb + write_kb) * 1024.0,
               "note": "traffic = 2 x FETCH_SIZE (gfx950 16-B/lane correction for the streaming reads) + WRITE_SIZE, KB -> bytes, per launch"
                       + ("; the layer kernel's weight slices (3 x 1.77 MB) are read once per XCD = 8 x, and its launches run at the mean "
                          "self-attention fill (150 cached keys)" if "layer" in kernel else "")}
        out = os.path.join(roofline.PROFILES, fname % rec["kernel_source_sha256_12"])
        with open(out, "w") as fh:
            json.dump(rec, fh, indent=1)
        print(out, json.dumps(rec))


if __name__ == "__main__":
    main()
