#!/bin/bash
# board power / shader clock under the fp32-class encoder forward (split GEMMs + split attention) and the fp16 one
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/power_r04; rm -rf $O; mkdir -p $O; cd $R
run() { name=$1; shift; env "$@" bash tools/power_sample2.sh $O/$name python3 tools/enc_only.py > $O/$name.log 2>&1; grep ENC $O/$name.log; }
run enc32_256x32 PROBE_SEQS=256 PROBE_TOKENS=32 RARC_ENC_PRECISION=fp32 PROBE_ITERS=400
run enc32_64x512 PROBE_SEQS=64 PROBE_TOKENS=512 RARC_ENC_PRECISION=fp32 PROBE_ITERS=100
run enc16_64x512 PROBE_SEQS=64 PROBE_TOKENS=512 RARC_ENC_PRECISION=fp16 PROBE_ITERS=300
python3 - $O <<'PY' | tee $R/gpurun_out/r04_power_encoder.txt
import json, sys, glob, os, statistics as st
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.rocmsmi"))):
    tup = []
    for line in open(f):
        line = line.strip()
        if not line.startswith("{"): continue
        try: d = json.loads(line)
        except Exception: continue
        c = d.get("card0", {})
        def num(key_part):
            for k, v in c.items():
                if key_part in k:
                    try: return float(str(v).strip("()Mhz W%").replace("Mhz", ""))
                    except Exception: pass
            return None
        s, p, u = num("sclk clock speed"), num("Socket Graphics Package Power") or num("Power (W)"), num("GPU use")
        if s is not None and p is not None and u is not None: tup.append((int(s), int(p), int(u)))
    busy = [t for t in tup if t[2] >= 99]
    name = os.path.basename(f)[:-8]
    if busy:
        print(f"== {name}: steady state ({len(busy)} samples, GPU use >= 99 %): sclk {min(b[0] for b in busy)}-{max(b[0] for b in busy)} MHz (median {int(st.median(b[0] for b in busy))}), "
              f"socket power {min(b[1] for b in busy)}-{max(b[1] for b in busy)} W (median {int(st.median(b[1] for b in busy))})")
    print("   all samples:", " ".join(f"({a},{b},{c})" for a, b, c in tup[:120]))
PY
