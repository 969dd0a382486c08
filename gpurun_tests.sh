cd $GRAFT_REPO_ROOT
ulimit -v 100000000
for lib in piqp_amd/lib/exp/libpiqp_amd_base.so piqp_amd/lib/libpiqp_amd.so; do
echo "== $lib"
PIQP_AMD_LIB=$PWD/$lib PIQP_AMD_DEBUG=trsv_ts timeout 300 python3 tools/prof_dense.py 4096 4096 0 1 0 2>&1 | grep "piqp_amd\]" | tail -2 | cut -c1-700
done
