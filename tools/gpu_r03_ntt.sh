#!/bin/bash
# stand-alone transform variants against the default library, interleaved: $1 = "tag tag ..."
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/ab_libs.sh "$1" --workload ntt --steps 20 --warmup 3 $2
