"""One-hop halo exchange for vertex-partitioned meshes (the reference has no multi-GPU code;
this is the build's own scaling path).

Every rank owns a contiguous range of vertices and all edges whose TARGET it owns, so stencil rows
never move.  Before a FieldConv the rank needs the feature rows of the remote SOURCES of those
edges (its halo): one all-to-all-v of (halo x C) complex rows over RCCL/xGMI (backend "nccl" on
ROCm; "gloo" in the CPU tests).  In the backward pass the transposed exchange returns the halo
rows' gradient contributions to their owners, where they are summed.  Row counts per peer are fixed
by the partition, so the plan (who sends which rows to whom) is built once per mesh.
"""
import torch
import torch.distributed as dist


class HaloPlan:
    """Built collectively.  `halo_global` lists the remote vertices this rank reads, grouped by
    owning rank (ascending global id works for contiguous ownership ranges `owner_bounds`)."""

    def __init__(self, n_owned, halo_global, owner_bounds, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_owned = int(n_owned)
        self.n_halo = int(halo_global.numel())
        bounds = owner_bounds.to(torch.int64).cpu()
        halo_global = halo_global.to(torch.int64).cpu()
        owner = torch.bucketize(halo_global, bounds[1:], right=True)
        if self.n_halo and not bool((owner[1:] >= owner[:-1]).all()):
            raise ValueError('halo_global must be grouped by owning rank')
        if self.n_halo and bool((owner == self.rank).any()):
            raise ValueError('halo_global contains vertices owned by this rank')
        self.recv_counts = torch.bincount(owner, minlength=self.world).tolist()
        # tell every owner how many (then which) of its rows this rank wants
        rc = torch.tensor(self.recv_counts, dtype=torch.int64, device=device)
        sc = torch.empty_like(rc)
        _all_to_all(sc, rc, None, None, group)
        self.send_counts = sc.tolist()
        want = halo_global.to(device)
        asked = torch.empty(sum(self.send_counts), dtype=torch.int64, device=device)
        _all_to_all(asked, want, self.send_counts, self.recv_counts, group)
        lo = int(bounds[self.rank])
        self.send_idx = (asked - lo).contiguous()              # local owned rows to ship, grouped by peer
        if self.send_idx.numel() and (int(self.send_idx.min()) < 0 or int(self.send_idx.max()) >= self.n_owned):
            raise ValueError('a peer asked for a vertex this rank does not own')


class _HaloExchange(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_owned, plan):
        ctx.plan = plan
        send = x_owned.index_select(0, plan.send_idx).contiguous()
        recv = torch.empty((plan.n_halo,) + tuple(x_owned.shape[1:]), dtype=x_owned.dtype, device=x_owned.device)
        _a2a(recv, send, plan.recv_counts, plan.send_counts, plan.group)
        return torch.cat((x_owned, recv), dim=0)

    @staticmethod
    def backward(ctx, g_local):
        plan = ctx.plan
        g_owned = g_local[: plan.n_owned].clone()
        send = g_local[plan.n_owned:].contiguous()
        recv = torch.empty((plan.send_idx.numel(),) + tuple(g_local.shape[1:]), dtype=g_local.dtype, device=g_local.device)
        _a2a(recv, send, plan.send_counts, plan.recv_counts, plan.group)
        g_owned.index_add_(0, plan.send_idx, recv)
        return g_owned, None


def _all_to_all(recv, send, recv_counts, send_counts, group):
    """all_to_all_single; device tensors are staged through the host when the group's backend is gloo (used to
    run several ranks on ONE GPU in tests -- RCCL refuses two ranks per device; production runs use "nccl")."""
    kw = {}
    if recv_counts is not None:
        kw = dict(output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts))
    if recv.is_cuda and dist.get_backend(group) == 'gloo':
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r, send.cpu(), group=group, **kw)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, group=group, **kw)


def _a2a(recv, send, recv_counts, send_counts, group):
    r = torch.view_as_real(recv) if recv.is_complex() else recv
    s = torch.view_as_real(send) if send.is_complex() else send
    _all_to_all(r, s, recv_counts, send_counts, group)


def halo_exchange(x_owned, plan):
    """(n_owned, C) -> (n_owned + n_halo, C): owned rows followed by the halo rows, differentiable."""
    return _HaloExchange.apply(x_owned, plan)
