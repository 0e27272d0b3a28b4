"""Multi-GPU layout of the MixQ linear: shard ROWS of W (output features) across ranks, one all-gather of the fp16
output -- and only when TP > 1 (BASELINE.json north_star).

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
The reference's only collective is an ``allreduce`` after an out_features//tp_size split (plugin.py:97,155-156),
which is shape-wrong for an N-split and guarded by ``assert tp_size == 1``; it is deliberately not reproduced.

Every tensor that is per-output-feature is sharded the same way: W int8 [N,K], sW [N], fp_weight [N,128] and the
decode ``qweight`` [K,N] (by column pairs).  ``fp_ind`` and the activations are replicated; the per-token
quantisation pre-pass is recomputed on every rank (HBM-bound, no communication).
"""
from typing import Dict, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_total: int, tp_size: int, rank: int, multiple: int = 16) -> Tuple[int, int]:
    """Contiguous [n0, n1) slice of the output features owned by ``rank``; N/tp must keep the operator's
    16-element alignment (all BASELINE shapes give multiples of 256)."""
    assert n_total % tp_size == 0, f"N={n_total} not divisible by tp_size={tp_size}"
    per = n_total // tp_size
    assert per % multiple == 0, f"N/tp={per} must be a multiple of {multiple}"
    return rank * per, (rank + 1) * per


def shard_packed(packed: Dict[str, np.ndarray], tp_size: int, rank: int) -> Dict[str, np.ndarray]:
    """Slice the 7-tensor contract of one layer (pack.pack_linear_weights output, true dtypes) for ``rank``.

    ``qweight`` is sliced in the interleaved image: output-feature pair j owns bytes [j*2K, (j+1)*2K), so a
    contiguous range of pairs is a contiguous byte range (see csrc/decode_kernels.hip)."""
    N, K = packed["weight"].shape
    n0, n1 = shard_bounds(N, tp_size, rank)
    out = dict(packed)
    out["weight"] = np.ascontiguousarray(packed["weight"][n0:n1])
    out["weights_scaling_factor"] = np.ascontiguousarray(packed["weights_scaling_factor"][n0:n1])
    out["fp_weight"] = np.ascontiguousarray(packed["fp_weight"][n0:n1])
    qw = np.ascontiguousarray(packed["qweight"]).reshape(-1)
    out["qweight"] = qw[n0 * K:n1 * K].reshape(K, n1 - n0).copy()
    if "scales" in packed:
        out["scales"] = np.ascontiguousarray(packed["scales"][n0:n1])
    if packed.get("bias") is not None:  # per output feature like sW: each rank adds its slice before the gather
        assert packed["bias"].shape == (N,), f"bias {packed['bias'].shape} vs N={N}"
        out["bias"] = np.ascontiguousarray(packed["bias"][n0:n1])
    return out


def _is_gloo(group) -> bool:
    try:
        return dist.get_backend(group) == "gloo"
    except Exception:  # noqa: BLE001
        return False


def all_gather_columns(x_local: torch.Tensor, group=None, tp_size: int = None) -> torch.Tensor:
    """[.., N/tp] per rank -> [.., N] on every rank with ONE collective (ncclAllGather under RCCL).

    RCCL moves contiguous buffers, so the gather lands rank-major ([tp, M, N/tp]); one strided copy puts the column
    blocks side by side (csrc/tp_kernels.hip's peer writes -- parallel.PeerGather -- avoid that pass where the ranks can
    map each other's output buffers).  Message per rank per call = M * N/tp * 2 bytes to each of the tp-1 peers.
    A gloo group (CPU tests, and the 2-ranks-on-one-GPU test) is served through host staging."""
    if tp_size is None:
        tp_size = dist.get_world_size(group)
    if tp_size == 1:
        return x_local
    lead = x_local.shape[:-1]
    n_loc = x_local.shape[-1]
    x2 = x_local.reshape(-1, n_loc).contiguous()
    dev = x2.device
    if x2.is_cuda and _is_gloo(group):
        x2 = x2.cpu()
    gathered = torch.empty((tp_size * x2.shape[0], n_loc), dtype=x2.dtype, device=x2.device)
    dist.all_gather_into_tensor(gathered, x2, group=group)  # rank-major concatenation along dim 0
    full = gathered.view(tp_size, x2.shape[0], n_loc).permute(1, 0, 2).reshape(x2.shape[0], tp_size * n_loc)
    return full.reshape(*lead, tp_size * n_loc).to(dev)
