"""librempeg_amd -- MI355X-native libswscale hot path (HIP kernels behind the sws_* C-ABI).

Python here is plumbing only: a ctypes binding of libswscale_hip.so plus helpers that use torch
for device memory, streams and torch.distributed.  The product is the shared library.
"""
from .swscale import (  # noqa: F401
    SwsContext, PIX_FMT, load_library, library_path, build_library,
    SWS_FAST_BILINEAR, SWS_BILINEAR, SWS_BICUBIC, SWS_X, SWS_POINT, SWS_AREA, SWS_BICUBLIN, SWS_GAUSS,
    SWS_SINC, SWS_LANCZOS, SWS_SPLINE, SWS_PRINT_INFO, SWS_FULL_CHR_H_INT, SWS_FULL_CHR_H_INP,
    SWS_ACCURATE_RND, SWS_BITEXACT, SWS_CS_ITU709, SWS_CS_ITU601, SWS_CS_DEFAULT, SWS_CS_BT2020,
    plane_layout, image_layout, DeviceFrame, HostFrame,
)
