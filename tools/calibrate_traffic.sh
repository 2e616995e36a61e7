#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the GPU box (one gpurun call): tools/traffic_calib.hip under two PMC passes.
#   tools/calibrate_traffic.sh ; results: gpurun_out/traffic_calib.txt
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/traffic_calib.hip -o /tmp/traffic_calib || exit 1
rm -rf gpurun_out/calib_fetch gpurun_out/calib_write
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_fetch -o calib -- /tmp/traffic_calib > gpurun_out/calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/calib_write -o calib -- /tmp/traffic_calib > gpurun_out/calib_write.log 2>&1
python3 - <<'PY' | tee gpurun_out/traffic_calib.txt
import csv, glob, collections
GiB = 1 << 30
# dispatch order within one repetition and what each one touches
plan = [("calib_read_u128", "read 1 GiB, 16 B per lane, coalesced", GiB, GiB),
        ("calib_read_u32", "read 1 GiB, 4 B per lane, coalesced", GiB, GiB),
        ("calib_read_u32", "read one word per 64 B (16 M words)", GiB // 16, GiB),
        ("calib_read_u32", "read one word per 128 B (8 M words)", GiB // 32, GiB),
        ("calib_read_u32_scattered", "read one word per 128 B line, random order", GiB // 32, GiB),
        ("calib_write_u32", "write 1 GiB, 4 B per lane, coalesced", GiB, GiB),
        ("calib_write_u32", "write one word per 128 B (8 M words)", GiB // 32, GiB),
        ("calib_write_u8_scattered", "write one byte per 128 B line, random order (8 M bytes)", GiB // 128, GiB)]
def rows(d, counter):
    out = []
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("calib_"):
                out.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], float(r["Counter_Value"])))
    return sorted(out)
print("| pattern | bytes touched | bytes of the lines touched (64 B lines for the 64 B stride, else 128 B) | reported KB x 1024 | reported / touched | reported / lines |")
print("|---|---|---|---|---|---|")
for counter, d in (("FETCH_SIZE", "calib_fetch"), ("WRITE_SIZE", "calib_write")):
    rs = rows(d, counter)
    per = len(plan)
    last = rs[-per:] if len(rs) >= per else rs       # third repetition
    for (kname, desc, touched, lines), (_, name, val) in zip(plan, last):
        if (counter == "FETCH_SIZE") != ("read" in desc):
            continue
        rep = val * 1024
        print("| %s: %s | %.0f MB | %.0f MB | %.0f MB | %.2f | %.2f |" % (counter, desc, touched / 1e6, lines / 1e6, rep / 1e6, rep / touched, rep / lines))
PY
