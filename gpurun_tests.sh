cd $GRAFT_REPO_ROOT
ulimit -v 100000000
timeout 600 python3 tools/chk_lib_variants.py piqp_amd/lib/libpiqp_amd.so piqp_amd/lib/libpiqp_amd.so@trsv_mfma 2>&1 | tail -32
