"""Stub `dgl` / `jax` / `jax_md` modules so that the *real* reference module
`/root/reference/code/nn_module.py` can be imported and executed on CPU in this
container (its third-party deps are not installable: no network).

TEST INFRASTRUCTURE ONLY.  Used by `oracle/make_golden.py` to generate the
golden vectors under `tests/golden/`.  Nothing in the product path
(`gamd_amd/`) may import this file, and it is never executed on the GPU box
(`/root/reference` does not exist there).

The stub implements the DGL 0.7.0 surface the reference touches, with the
semantics listed in SURVEY.md §8(c):

* `dgl.graph((src, dst))`            -> edge list container, num_nodes = max id + 1
* `.edges()`                         -> (src, dst) in insertion order
* `.edata/.ndata/.srcdata/.dstdata`  -> plain dicts (ndata/srcdata/dstdata alias)
* `.local_scope()`                   -> context manager restoring the dicts
* `.add_self_loop()`                 -> returns a NEW graph, receiver untouched
                                        (functional alias in DGL >= 0.5; call sites
                                        nn_module.py:364,518,652 discard the result).
                                        `INPLACE_SELF_LOOP = True` switches the stub to the
                                        OTHER reading (DGL < 0.5: the receiver itself gets
                                        one i -> i edge per node, edge data zero-filled) so
                                        that the build's self_loop_mode 1 has reference
                                        outputs to be checked against
* `.update_all(fn.src_mul_edge(a,b,m), fn.sum(m,o))`
                                     -> ndata[o][v] = sum_{u->v} ndata[a][u] * edata[b][uv]
                                        (zeros for nodes without in-edges)
* `dgl.add_reverse_edges`, `.has_edges_between(u, v)`, `dgl.batch`
"""
import sys
import types
import contextlib

import torch


# False: DGL >= 0.5 functional add_self_loop (the pinned DGL 0.7.0); True: in-place add_self_loop of DGL < 0.5
INPLACE_SELF_LOOP = False


class _StubGraph:
    is_block = False

    def __init__(self, src, dst, num_nodes=None):
        self._src = torch.as_tensor(src).long()
        self._dst = torch.as_tensor(dst).long()
        if num_nodes is None:
            num_nodes = 0
            if self._src.numel():
                num_nodes = int(max(self._src.max(), self._dst.max())) + 1
        self._n = num_nodes
        self.edata = {}
        self.ndata = {}
        self.srcdata = self.ndata
        self.dstdata = self.ndata

    def edges(self):
        return self._src, self._dst

    def num_nodes(self):
        return self._n

    number_of_nodes = num_nodes

    def number_of_dst_nodes(self):
        return self._n

    def num_edges(self):
        return int(self._src.numel())

    @contextlib.contextmanager
    def local_scope(self):
        e, n = dict(self.edata), dict(self.ndata)
        try:
            yield
        finally:
            self.edata.clear(); self.edata.update(e)
            self.ndata.clear(); self.ndata.update(n)

    def add_self_loop(self):
        loops = torch.arange(self._n)
        g = _StubGraph(torch.cat([self._src, loops]), torch.cat([self._dst, loops]), self._n)
        for k, v in self.edata.items():
            pad = torch.zeros((self._n,) + tuple(v.shape[1:]), dtype=v.dtype)
            g.edata[k] = torch.cat([v, pad])
        if INPLACE_SELF_LOOP:
            self._src, self._dst = g._src, g._dst
            self.edata.clear()
            self.edata.update(g.edata)
            return self
        return g

    def update_all(self, msg, red):
        kind, a, b, m = msg
        rkind, m2, out = red
        assert kind == 'u_mul_e' and rkind == 'sum' and m == m2
        mval = self.ndata[a][self._src] * self.edata[b]
        res = torch.zeros((self._n,) + tuple(mval.shape[1:]), dtype=mval.dtype)
        res.index_add_(0, self._dst, mval)
        self.ndata[out] = res

    def has_edges_between(self, u, v):
        n = max(self._n, int(torch.as_tensor(u).max()) + 1, int(torch.as_tensor(v).max()) + 1)
        key = self._src * n + self._dst
        q = torch.as_tensor(u).long() * n + torch.as_tensor(v).long()
        return torch.isin(q, key)


def _graph(data, num_nodes=None, **kw):
    return _StubGraph(data[0], data[1], num_nodes)


def _add_reverse_edges(g):
    return _StubGraph(torch.cat([g._src, g._dst]), torch.cat([g._dst, g._src]), g._n)


def _batch(graphs):
    off, srcs, dsts = 0, [], []
    for g in graphs:
        srcs.append(g._src + off); dsts.append(g._dst + off); off += g._n
    out = _StubGraph(torch.cat(srcs), torch.cat(dsts), off)
    for k in graphs[0].edata:
        out.edata[k] = torch.cat([g.edata[k] for g in graphs])
    return out


def install():
    """Insert the stub modules into sys.modules (idempotent)."""
    if 'dgl' in sys.modules and getattr(sys.modules['dgl'], '_gamd_stub', False):
        return
    dgl = types.ModuleType('dgl')
    dgl._gamd_stub = True
    dgl.DGLGraph = _StubGraph
    dgl.graph = _graph
    dgl.add_reverse_edges = _add_reverse_edges
    dgl.batch = _batch
    fn = types.ModuleType('dgl.function')
    fn.src_mul_edge = lambda a, b, m: ('u_mul_e', a, b, m)
    fn.u_mul_e = fn.src_mul_edge
    fn.sum = lambda m, o: ('sum', m, o)
    ops = types.ModuleType('dgl.ops')
    ops.edge_softmax = None
    utils = types.ModuleType('dgl.utils')
    utils.expand_as_pair = None
    nn = types.ModuleType('dgl.nn')
    dgl.function, dgl.ops, dgl.utils, dgl.nn = fn, ops, utils, nn
    sys.modules.update({'dgl': dgl, 'dgl.function': fn, 'dgl.ops': ops,
                        'dgl.utils': utils, 'dgl.nn': nn})

    jax = types.ModuleType('jax')
    jax.jit = lambda f=None, **kw: f if f is not None else (lambda g: g)
    jax.vmap = lambda f, *a, **k: f
    jnp = types.ModuleType('jax.numpy')
    jax.numpy = jnp
    jax_md = types.ModuleType('jax_md')
    space = types.ModuleType('jax_md.space')
    space.pairwise_displacement = None
    partition = types.ModuleType('jax_md.partition')
    jax_md.space, jax_md.partition = space, partition
    sys.modules.update({'jax': jax, 'jax.numpy': jnp, 'jax_md': jax_md,
                        'jax_md.space': space, 'jax_md.partition': partition})


def import_reference(ref_root='/root/reference'):
    """Import the reference's nn_module / md_module (pure torch once stubbed)."""
    import os
    install()
    code = os.path.join(ref_root, 'code')
    if code not in sys.path:
        sys.path.insert(0, code)
    import nn_module  # noqa: E402  (the reference's own file, read in place)
    import md_module  # noqa: E402
    return nn_module, md_module
