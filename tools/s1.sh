cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s12
timeout 600 python tools/multi_stream_bs1.py res101_coco 8 > gpurun_out/s12/ms8.txt 2>&1
timeout 600 python tools/multi_stream_bs1.py res101_coco 2 > gpurun_out/s12/ms2.txt 2>&1
