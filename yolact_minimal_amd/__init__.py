"""yolact_minimal_amd — MI355X-native (gfx950) hot path of YOLACT.

Only what the data-parallel hot path needs lives here:
  csrc/      hand-written HIP kernels + the C-ABI (`include/yolact_hip.h`)
  hip.py     ctypes binding of that C-ABI (fails loudly if the .so is missing)
  engine.py  layer plan: packs weights, owns HBM buffers, launches the kernels
  modules/   `Yolact` with the reference's constructor / forward / state-dict surface
  utils/     `nms`, `after_nms`, `make_anchors` with the reference's signatures
  config.py  cfg classes + `get_config`
"""
__all__ = ['config', 'hip', 'engine', 'modules', 'utils']
