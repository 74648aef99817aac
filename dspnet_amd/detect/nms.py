"""Pixel-coordinate NMS front end (counterpart of detect/nms.py of the reference).

Same callables -- `nms(dets, thresh)`, `cpu_nms`, `gpu_nms` and the three `*_wrapper` factories
(detect/nms.py:6-21) -- all backed by the HIP kernels behind include/dspn_nms.h; there is no CPU path.
`dets`: (n, 5) [x1, y1, x2, y2, score], numpy array or torch tensor; returns the list of kept indices in
descending score order, like the reference.  `nms` / `gpu_nms` drop a box whose overlap with a kept one is
> thresh (detect/nms.py:55, cython/nms_kernel.cu:68), `cpu_nms` when it is >= thresh (cython/cpu_nms.pyx:65).
`bbox_overlaps` / `bbox_overlaps_cython`: the (N, K) float64 overlap matrix of cython/bbox.pyx:15-55."""
import ctypes

import numpy as np
import torch

from .. import _lib

_c = ctypes
_lib.register({
    "dspn_nms_pixel_workspace_bytes": (_c.c_size_t, [_c.c_int]),
    "dspn_nms_pixel_f32": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_float, _c.c_int, _c.c_void_p, _c.c_void_p,
                                      _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "dspn_bbox_overlaps_f64": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p]),
})


def bbox_overlaps(boxes, query_boxes, device=None):
    """cython/bbox.pyx:15-55 `bbox_overlaps_cython(boxes (N,4), query_boxes (K,4)) -> (N,K)`, float64, on the GPU.
    numpy arrays in -> numpy array out (the reference's contract); torch CUDA tensors in -> a CUDA tensor out."""
    as_numpy = isinstance(boxes, np.ndarray) or isinstance(query_boxes, np.ndarray)
    dev = device or torch.device("cuda", torch.cuda.current_device())

    def dev64(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)) if isinstance(a, np.ndarray) else a.detach().to(torch.float64)
        if t.ndim != 2 or t.shape[1] != 4:
            raise _lib.DspnError("bbox_overlaps: boxes must be (n, 4) [x1, y1, x2, y2]")
        return (t if t.is_cuda else t.to(dev)).contiguous()

    b, q = dev64(boxes), dev64(query_boxes)
    out = torch.zeros(b.shape[0], q.shape[0], dtype=torch.float64, device=b.device)
    _lib.check(_lib.lib().dspn_bbox_overlaps_f64(b.data_ptr(), b.shape[0], q.data_ptr(), q.shape[0], out.data_ptr(),
                                                 torch.cuda.current_stream(b.device).cuda_stream), "bbox_overlaps")
    return out.cpu().numpy() if as_numpy else out


bbox_overlaps_cython = bbox_overlaps      # the reference's name (cython/bbox.pyx:15)


def _run(dets, thresh, suppress_ge, device=None):
    if isinstance(dets, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(dets, dtype=np.float32))
    else:
        t = dets.detach().to(torch.float32)
    if t.ndim != 2 or t.shape[1] != 5:
        raise _lib.DspnError("nms: dets must be (n, 5) [x1, y1, x2, y2, score]")
    if not t.is_cuda:
        t = t.to(device or torch.device("cuda", torch.cuda.current_device()))
    t = t.contiguous()
    n = t.shape[0]
    if n == 0:
        return []
    lib = _lib.lib()
    keep = torch.empty(n, dtype=torch.int32, device=t.device)
    num = torch.zeros(1, dtype=torch.int32, device=t.device)
    ws = torch.empty(max(int(lib.dspn_nms_pixel_workspace_bytes(n)), 16), dtype=torch.uint8, device=t.device)
    _lib.check(lib.dspn_nms_pixel_f32(t.data_ptr(), n, float(thresh), int(suppress_ge), keep.data_ptr(), num.data_ptr(),
                                      ws.data_ptr(), ws.numel(), torch.cuda.current_stream(t.device).cuda_stream),
               "nms_pixel")
    return keep[:int(num.item())].cpu().tolist()


def nms(dets, thresh):
    """detect/nms.py:24-58: keep while overlap <= thresh"""
    return _run(dets, thresh, 0)


def gpu_nms(dets, thresh, device_id=0):
    """cython/gpu_nms.pyx + nms_kernel.cu: suppress overlap > thresh (a NaN overlap survives, unlike in `nms`)"""
    return _run(dets, thresh, 2, torch.device("cuda", device_id))


def cpu_nms(dets, thresh):
    """cython/cpu_nms.pyx:17-68 semantics (suppress overlap >= thresh); runs on the GPU like the others"""
    return _run(dets, thresh, 1)


def py_nms_wrapper(thresh):
    return lambda dets: nms(dets, thresh)


def cpu_nms_wrapper(thresh):
    return lambda dets: cpu_nms(dets, thresh)


def gpu_nms_wrapper(thresh, device_id):
    return lambda dets: gpu_nms(dets, thresh, device_id)
