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
        self._early = {}           # gradient exchanges started by overlap_backward's hook: data_ptr -> (work, send, recv)
        # per-call constants of the two exchanges (the partitioned step is bound by the host's enqueue rate: every
        # microsecond of Python per collective counts)
        self._staged = dist.get_backend(group) == 'gloo'
        self._kw_fwd = dict(output_split_sizes=list(self.recv_counts), input_split_sizes=list(self.send_counts))
        self._kw_bwd = dict(output_split_sizes=list(self.send_counts), input_split_sizes=list(self.recv_counts))
        self._pending = None       # forward exchange still in flight (halo_exchange(..., deferred=True)): (work, send, x_local)

    def wait_forward(self):
        """The current stream waits for the forward exchange started by halo_exchange(..., deferred=True), if any."""
        pending, self._pending = self._pending, None
        if pending is not None and pending[0] is not None:
            pending[0].wait()


class _HaloExchange(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_owned, plan, deferred):
        ctx.plan = plan
        plan.wait_forward()                                     # an earlier deferred exchange nobody waited for
        send = x_owned.index_select(0, plan.send_idx).contiguous()
        x_local = torch.empty((plan.n_owned + plan.n_halo,) + tuple(x_owned.shape[1:]), dtype=x_owned.dtype, device=x_owned.device)
        recv = x_local[plan.n_owned:]                           # the halo rows arrive in place: no concatenation afterwards
        work = _exchange(plan, recv, send, True, async_op=deferred)
        x_local[: plan.n_owned].copy_(x_owned)
        if deferred:
            plan._pending = (work, send, x_local)
        return x_local

    @staticmethod
    def backward(ctx, g_local):
        plan = ctx.plan
        g_owned = g_local[: plan.n_owned].clone()
        early = plan._early.pop(g_local.data_ptr(), None)
        plan._early.clear()
        if early is not None and early[2].shape[1:] == g_local.shape[1:]:
            work, _, recv = early                       # started between the convolution's two backward kernels
            if work is not None:
                work.wait()
        else:
            send = g_local[plan.n_owned:].contiguous()
            recv = torch.empty((plan.send_idx.numel(),) + tuple(g_local.shape[1:]), dtype=g_local.dtype, device=g_local.device)
            _exchange(plan, recv, send, False)
        g_owned.index_add_(0, plan.send_idx, recv)
        return g_owned, None, None


def _all_to_all(recv, send, recv_counts, send_counts, group, async_op=False):
    """all_to_all_single; device tensors are staged through the host when the group's backend is gloo (used to
    run several ranks on ONE GPU in tests -- RCCL refuses two ranks per device; production runs use "nccl").
    async_op: returns the Work handle (None on the host-staged path, which completes here)."""
    kw = {}
    if recv_counts is not None:
        kw = dict(output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts))
    if recv.is_cuda and dist.get_backend(group) == 'gloo':
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r, send.cpu(), group=group, **kw)
        recv.copy_(r)
        return None
    return dist.all_to_all_single(recv, send, group=group, async_op=async_op, **kw)


def _exchange(plan, recv, send, forward, async_op=False):
    """The plan's exchange of complex rows (forward: owners -> halos; else halos -> owners)."""
    r = torch.view_as_real(recv) if recv.is_complex() else recv
    s = torch.view_as_real(send) if send.is_complex() else send
    kw = plan._kw_fwd if forward else plan._kw_bwd
    if plan._staged and recv.is_cuda:
        h = torch.empty(r.shape, dtype=r.dtype)
        dist.all_to_all_single(h, s.cpu(), group=plan.group, **kw)
        r.copy_(h)
        return None
    return dist.all_to_all_single(r, s, group=plan.group, async_op=async_op, **kw)


def overlap_backward(graph, plan):
    """Hide the gradient halo exchange under the filter-gradient kernel.  The backward pass of a convolution is two
    kernels: the first completes gx (including the rows of the halo vertices, which belong to their owners), the second
    only reads the slabs the first left behind.  With this hook on the mesh's SupportGraph the transposed exchange of
    the halo rows is enqueued between the two (asynchronously, on RCCL's stream) and halo_exchange's backward node later
    only waits for it.  `graph`: a view of the mesh's graph (SupportGraph.view()) or a graph of the caller's own."""
    graph._own('overlap_backward')
    def on_gx(gx):
        if gx.shape[0] != plan.n_owned + plan.n_halo:
            return
        send = gx[plan.n_owned:]                                   # contiguous rows of a contiguous tensor
        recv = torch.empty((plan.send_idx.numel(),) + tuple(gx.shape[1:]), dtype=gx.dtype, device=gx.device)
        work = _exchange(plan, recv, send, False, async_op=True)
        plan._early.clear()
        plan._early[gx.data_ptr()] = (work, send, recv)
    graph.on_gx = on_gx


def overlap_forward(graph, plan, n_interior, whole_rounds=True):
    """Hide the forward halo exchange under the convolution of the interior targets.  With the owned vertices numbered so
    that the first `n_interior` read owned sources only (data.sphere_partition(..., interior_first=True)), the forward pass
    over this graph becomes two launches: targets [0, n_interior) while the halo rows are still arriving, then -- after the
    stream waited for the exchange -- the boundary targets.  Use with halo_exchange(x, plan, deferred=True); per mesh."""
    graph._own('overlap_forward')
    n_first = int(n_interior)
    if not 0 <= n_first <= plan.n_owned:
        raise ValueError('n_interior must lie in [0, n_owned]')
    if whole_rounds and graph.rowptr_t.is_cuda:
        # The ring-major forward kernel is a persistent grid of two workgroups per CU, 16 targets per tile: a first launch
        # of whole rounds leaves no thinly filled last round behind (config 2 on one MI355X: +11 us for the second launch
        # when split at 16 384 = 2 rounds, +40 us when split at the 18 504 interior targets; tools/split_forward_cost.py).
        # Any prefix of the interior targets is interior.
        per_round = 16 * 2 * torch.cuda.get_device_properties(graph.rowptr_t.device).multi_processor_count
        if n_first >= per_round:
            n_first -= n_first % per_round
    graph.forward_split = (n_first, plan.wait_forward)


def halo_exchange(x_owned, plan, deferred=False):
    """(n_owned, C) -> (n_owned + n_halo, C): owned rows followed by the halo rows, differentiable.

    deferred: return while the exchange is still in flight on the communicator's stream.  The halo rows of the result must
    then not be read before plan.wait_forward(); a convolution over a graph prepared with overlap_forward(graph, plan, ...)
    does that between its interior and boundary launches.  For any other consumer leave it False."""
    return _HaloExchange.apply(x_owned, plan, bool(deferred))
