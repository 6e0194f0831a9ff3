cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s19
timeout 120 tools/micro/mfma_lds > gpurun_out/s19/mfma_lds.txt 2>&1
