#!/bin/bash
# round 5, first GPU session (prepared at the end of round 4): the queued variants of experiments/r05/ -- built in the container first:
#   for v in "fp64md fp64_moddown" "grid4 grid4" "gridhead grid_head"; do set -- $v; bash tools/build_patched_variant.sh $1 experiments/r05/$2.patch; done
# (1) the whole GPU suite under the FP64 mod-down variant (its stand-alone ks_tail_kernel has never run on a GPU), (2) interleaved A/B of the three
# variants on the headline configuration, (3) the mod-down variant on n = 16384, chi_sq and dot_prod (13 rotations per program: ks_tail proper).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s1; mkdir -p $O
V=$GRAFT_REPO_ROOT/sunscreen_amd/lib/variants
( time HIPBFV_LIB=$V/libhipbfv_fp64md.so timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_fp64md.log 2>&1; tail -3 $O/pytest_fp64md.log
bash tools/ab_libs.sh "fp64md grid4 gridhead" --steps 5 --warmup 2 --repeats 3 > $O/ab_n8192.txt 2>&1; cat $O/ab_n8192.txt
bash tools/ab_libs.sh "fp64md" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 > $O/ab_n16384.txt 2>&1; cat $O/ab_n16384.txt
bash tools/ab_libs.sh "fp64md" --workload chi_sq --n 16384 --batch 256 --steps 5 --warmup 2 --repeats 3 > $O/ab_chi.txt 2>&1; cat $O/ab_chi.txt
bash tools/ab_libs.sh "fp64md" --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 --repeats 3 > $O/ab_dot.txt 2>&1; cat $O/ab_dot.txt
