"""torch_cluster 1.6.0 ``radius`` / ``radius_graph`` and torch_scatter 2.1.0
``scatter`` semantics, restated brute-force on CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Both wheels are un-vendored
third-party dependencies of the reference (env.yaml:29-30) => **parity
unpinned**; the semantics restated here are the published ones of the GPU
kernels the reference runs on:

* ``radius(x, y, r, batch_x, batch_y, max_num_neighbors)``: for every query
  ``y[i]`` the points ``x[j]`` of the same batch element with
  ``|x_j - y_i|^2 < r^2`` (strict), scanned in index order, at most
  ``max_num_neighbors`` kept ("first K by index" = the CUDA kernel's order).
  Returns ``[2, E]`` with row 0 = query index, row 1 = point index.
* ``radius_graph(x, r, batch, loop=False, max_num_neighbors=32)``
  (flow ``source_to_target``): ``radius(x, x, r, ..., max_num_neighbors + 1)``,
  rows swapped so row 0 = neighbour, row 1 = centre, self loops removed.
* ``scatter(src, index, dim=0, dim_size, reduce)``: ``sum`` via index_add_,
  ``mean`` divides by ``count.clamp(min=1)``.

Call sites: tpscore.py:586,613,655-660,721-723,747-749 (radius*),
tpscore.py:190, conformer_utils.py:442 (scatter).
"""
import torch


def radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32):
    if batch_x is None:
        batch_x = x.new_zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = y.new_zeros(y.shape[0], dtype=torch.long)
    rows, cols = [], []
    if y.shape[0] == 0 or x.shape[0] == 0:
        return torch.zeros(2, 0, dtype=torch.long)
    nb = int(max(batch_x.max().item(), batch_y.max().item())) + 1
    r2 = float(r) * float(r)
    x_idx_all = torch.arange(x.shape[0])
    y_idx_all = torch.arange(y.shape[0])
    for b in range(nb):
        xm, ym = batch_x == b, batch_y == b
        xi, yi = x_idx_all[xm], y_idx_all[ym]
        if xi.numel() == 0 or yi.numel() == 0:
            continue
        d = y[yi][:, None, :] - x[xi][None, :, :]
        d2 = (d * d).sum(-1)
        m = d2 < r2
        # keep the first max_num_neighbors hits per query in index order
        rank = torch.cumsum(m.to(torch.long), dim=1)
        m = m & (rank <= max_num_neighbors)
        q, p = m.nonzero(as_tuple=True)
        rows.append(yi[q])
        cols.append(xi[p])
    if not rows:
        return torch.zeros(2, 0, dtype=torch.long)
    return torch.stack([torch.cat(rows), torch.cat(cols)], dim=0)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow="source_to_target"):
    assert flow in ("source_to_target", "target_to_source")
    ei = radius(x, x, r, batch, batch, max_num_neighbors if loop else max_num_neighbors + 1)
    if flow == "source_to_target":
        row, col = ei[1], ei[0]
    else:
        row, col = ei[0], ei[1]
    if not loop:
        m = row != col
        row, col = row[m], col[m]
    return torch.stack([row, col], dim=0)


def scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    dim_size = int(dim_size)
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    out.index_add_(0, index, src)
    if reduce in ("sum", "add"):
        return out
    assert reduce == "mean"
    cnt = torch.zeros(dim_size, dtype=src.dtype)
    cnt.index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype))
    cnt = cnt.clamp(min=1)
    return out / cnt.view((-1,) + (1,) * (src.dim() - 1))


def scatter_add(src, index, dim=0, dim_size=None):
    return scatter(src, index, dim, dim_size, "sum")


def scatter_mean(src, index, dim=0, dim_size=None):
    return scatter(src, index, dim, dim_size, "mean")
