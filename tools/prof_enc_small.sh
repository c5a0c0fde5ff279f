#!/bin/bash
# is the encoder's fixed cost per forward launch overhead?  sum of kernel time (rocprofv3) against wall time, 32 and 256 sequences
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_enc_small; rm -rf $O; mkdir -p $O; cd $R
for P in fp32 fp16; do for S in 32 256; do
  export PROBE_SEQS=$S RARC_ENC_PRECISION=$P
  python3 tools/enc_only.py 2>/dev/null | grep ENC
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$P$S -- python3 tools/enc_only.py > $O/$P$S.log 2>&1
  f=$(ls -t $O/$P$S/*/*kernel_stats.csv | head -1); cp $f $R/gpurun_out/r04_enc_${P}_${S}x32_kernel_stats.csv
  python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = [r for r in rows if "rarc" in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in k); calls = sum(int(r["Calls"]) for r in k)
print(f"   kernels: {tot / 6 / 1e6:.3f} ms per forward in {calls // 6} launches")
for r in sorted(k, key=lambda r: -int(r["TotalDurationNs"]))[:8]:
    print("     ", r["Name"][:70].ljust(70), int(r["Calls"]) // 6, f'{float(r["AverageNs"]) / 1e3:.1f} us')
PY
done; done
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete
