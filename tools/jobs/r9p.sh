/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
