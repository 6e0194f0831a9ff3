mkdir -p gpurun_out/r9q
python tools/cu_partition_probe.py --parts 4 --out gpurun_out/r9q/rows_q4.json 2>gpurun_out/r9q/err4.txt | tail -4
python tools/cu_partition_probe.py --parts 2 --out gpurun_out/r9q/rows_q2.json 2>gpurun_out/r9q/err2.txt | tail -4
tail -3 gpurun_out/r9q/err4.txt
