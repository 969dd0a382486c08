#!/bin/bash
# rocprofv3 kernel stats of the n = 500k chain (tools/prof_multistage.py --c5) on the GPU box; prints the sweep / factor kernels.
# usage: bash tools/run_c5prof.sh <tag>   (PIQP_AMD_* environment passes through; output under gpurun_out/prof_c5_<tag>)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_c5_$1 -o c5 -- python3 $GRAFT_REPO_ROOT/tools/prof_multistage.py --c5 --reps 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
echo "== $1"; python3 tools/rocprof_summary.py gpurun_out/prof_c5_$1 2>&1 | grep "wave<\|top_factor\|subtree_factor\|front_factor" | cut -c1-120
