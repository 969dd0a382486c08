cd $GRAFT_REPO_ROOT
ulimit -v 100000000
for t in "" multi_update=2; do
echo "== $t"
PIQP_AMD_DEBUG=$t timeout 600 python3 tools/prof_sparse.py --spread 300 --row-nnz 10 --no-oracle --reps 5 2>&1 | grep "^device: factor" | cut -c1-200
PIQP_AMD_DEBUG=$t timeout 600 python3 tools/prof_sparse.py --spread 1500 --row-nnz 10 --no-oracle --reps 2 2>&1 | grep "^device: factor" | cut -c1-200
done
