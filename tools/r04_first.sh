#!/bin/bash
# round 4, first GPU contact: box facts, the persistence tests, the whole GPU suite, same-box A/B of the LM forward
# (round-3 encoder.hip with the spilling residual epilogues vs this tree)
mkdir -p gpurun_out
{
  echo "== box"; nproc; free -g | head -2; df -h /tmp /dev/shm . 2>/dev/null; mount | grep -E " /tmp | /dev/shm " ; lscpu | grep -E "Model name|Socket|Thread|Core" 
} > gpurun_out/r04_box.txt 2>&1
python -m pytest tests/test_gpu_persistence.py -x -q -s > gpurun_out/r04_persist_tests.log 2>&1
echo "persist rc=$?" >> gpurun_out/r04_persist_tests.log
python -m pytest tests -m gpu -x -q > gpurun_out/r04_gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r04_gpu_suite.log
for i in 1 2 3; do
  RARC_LIBRARY=$PWD/rag-arc_amd/lib/librarc_hip_r3enc.so PROBE_REPS=5 python tools/lm_only.py | sed 's/^/r3enc: /'
  PROBE_REPS=5 python tools/lm_only.py | sed 's/^/r04:   /'
done > gpurun_out/r04_lm_ab.txt 2>&1
tail -5 gpurun_out/r04_persist_tests.log; tail -3 gpurun_out/r04_gpu_suite.log; cat gpurun_out/r04_lm_ab.txt; cat gpurun_out/r04_box.txt
