#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
python3 -m pytest tests/test_gpu_adversarial.py tests/test_gpu_persistence.py -q -x 2>&1 | tail -5
