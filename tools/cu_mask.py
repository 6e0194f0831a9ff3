"""HIP streams confined to a share of the compute units (`hipExtStreamCreateWithCUMask`), for requests in flight that should not
compete for the same CUs.

Measured on the MI355X (tools/micro/cu_mask_probe.hip): bit i of the mask is CU i // 8 of XCD i % 8, a mask has to leave every XCD
at least one CU (one that does not is ignored: the launch runs on all 256), so a share is "the same CUs of every XCD" — e.g. CUs
0..7 of each XCD = bits 0..63 = a quarter of the chip that still reaches all eight L2s.  A hipGraph captured on a masked stream and
launched into it keeps the mask; two streams with disjoint masks run side by side at full speed (22.6 ms alone, 22.8 ms together).
The runtime call is made through ctypes on the libamdhip64 torch has loaded; the stream is handed to torch as an ExternalStream."""
import ctypes

import torch

_lib = None
_keep = []          # (the runtime owns the streams for the life of the process: they are never destroyed)

CUS_PER_XCD = 32
XCDS = 8


def _hip():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL('libamdhip64.so')
        _lib.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        _lib.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
    return _lib


def share_mask(part, parts):
    """256-bit mask (8 x uint32) of share `part` of `parts`: CUs [part * 32 / parts, (part + 1) * 32 / parts) of every XCD."""
    if not (0 < parts <= CUS_PER_XCD and 0 <= part < parts and CUS_PER_XCD % parts == 0):
        raise ValueError(f'share {part} of {parts}: parts must divide {CUS_PER_XCD}')
    per = CUS_PER_XCD // parts
    words = [0] * 8
    for cu in range(part * per, (part + 1) * per):
        for xcd in range(XCDS):
            bit = cu * XCDS + xcd
            words[bit // 32] |= 1 << (bit % 32)
    return words


def masked_stream(device, part, parts):
    """A torch stream whose launches run on share `part` of `parts` of the CUs (see share_mask)."""
    device = torch.device(device)
    words = (ctypes.c_uint32 * 8)(*share_mask(part, parts))
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = _hip().hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0 or not st.value:
        raise RuntimeError(f'hipExtStreamCreateWithCUMask failed (rc={rc})')
    _keep.append(st)
    return torch.cuda.ExternalStream(st.value, device=device)
