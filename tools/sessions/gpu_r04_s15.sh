#!/bin/bash
# round 4, GPU session 15: the whole GPU suite after the alpha_sk fix of the mixed tail (behzcore.hpp), then the suites of the switches
# whose runs had caught it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s15; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
