cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_mma
python $R/tools/conv_layer_bench.py --cfgs 128x128:1:22:0,128x128:1:0:3,128x128:1:3:3,128x128:1:0:6,128x64:1:0:3,64x128:1:0:3,64x64:1:0:3,128x64:1:0:6 2>&1 | grep -v amdgpu > $R/gpurun_out/pmc_mma/layer_times2.log
python $R/tools/conv_layer_bench.py --shape 8,34,34,1024,256,1,1 --cfgs 64x64:1:22:0,64x64:1:0:3,128x64:1:0:3,64x64:1:0:6 2>&1 | grep -v amdgpu >> $R/gpurun_out/pmc_mma/layer_times2.log
python $R/tools/conv_layer_bench.py --shape 8,34,34,256,256,3,1 --cfgs 64x64:1:22:0,64x64:1:0:3,128x64:1:0:3,64x64:1:0:6 2>&1 | grep -v amdgpu >> $R/gpurun_out/pmc_mma/layer_times2.log
python $R/tools/conv_layer_bench.py --shape 8,136,136,64,256,1,1 --cfgs 128x128:1:22:0,128x128:1:0:3,128x64:1:0:3 2>&1 | grep -v amdgpu >> $R/gpurun_out/pmc_mma/layer_times2.log
python $R/tools/race_probe.py 300 2>&1 | grep -v amdgpu | grep launches >> $R/gpurun_out/pmc_mma/layer_times2.log
cat $R/gpurun_out/pmc_mma/layer_times2.log
