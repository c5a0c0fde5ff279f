#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python3 -m pytest tests -m gpu -q > $O/r04_final_pytest.log 2>&1; echo "pytest exit $?" >> $O/r04_final_pytest.log
tail -8 $O/r04_final_pytest.log
bash tools/prof_enc_small.sh 2>&1 | tee $O/r04_enc_small.txt
