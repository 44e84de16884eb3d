"""Set-up utilities of libmpc++ on the device (reference include/mpc/Utils.hpp)."""
import ctypes as C

from . import _capi
from ._capi import check


def discretization(A, B, Ts, device=0, stream=None):
    """mpc::discretization (Utils.hpp:23-47) for a batch: A [Bn, nx, nx], B [Bn, nx, nu] (row-major tensors as usual in
    torch), Ts a float or a [Bn] tensor.  Returns (Ad, Bd) on the device.  A disturbance matrix Be (Utils.hpp:63-89)
    is discretised by concatenating it to B's columns."""
    import torch
    dev = torch.device("cuda", device)
    A = torch.as_tensor(A, dtype=torch.float64).to(dev); B = torch.as_tensor(B, dtype=torch.float64).to(dev)
    if A.dim() == 2:
        A, B = A[None], B[None]
    n, nx, nu = A.shape[0], A.shape[1], B.shape[2]
    per = torch.is_tensor(Ts) and Ts.numel() > 1
    ts = (Ts.to(dev, torch.float64).contiguous() if per else torch.full((1,), float(Ts), dtype=torch.float64, device=dev))
    # the C ABI takes Eigen's column-major layout: transpose the last two axes
    Ac = A.transpose(1, 2).contiguous(); Bc = B.transpose(1, 2).contiguous()
    Ad = torch.empty_like(Ac); Bd = torch.empty_like(Bc)
    s = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    check(_capi.lib().mpcx_discretize_batch(device, nx, nu, n, Ac.data_ptr(), Bc.data_ptr(), ts.data_ptr(), int(per),
                                            Ad.data_ptr(), Bd.data_ptr(), s))
    return Ad.transpose(1, 2), Bd.transpose(1, 2)
