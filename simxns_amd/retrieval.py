"""Exhaustive inner-product index on the MI355X -- the role faiss.IndexFlatIP + index_cpu_to_all_gpus(shard=True) play
in the reference's generate job (SimANS/co_training/co_training_generate.py:359-384, 415-421).

One process per GPU holds a contiguous shard of the corpus embeddings in HBM (8.8 M x 768 f32 = 27 GB: the whole
MS-MARCO corpus fits one 288 GB GPU; with W ranks each holds 1/W).  search() = local top-k by the HIP kernels, then --
when a process group is given -- an all_gather of the W x [nq,k] candidates over RCCL and the same top-k kernel as the
merge.  No CPU path: without the HIP library every call raises."""
import ctypes as C

import torch

from . import _lib as L


class FlatIPIndex(object):
    """index = FlatIPIndex(dim); index.add(emb); D, I = index.search(q, k)   (faiss.IndexFlatIP surface)."""

    def __init__(self, dim, id_base=0, chunk=65536, query_block=8192):
        """``chunk``: corpus rows scored per pass; ``query_block``: queries per pass -- the score workspace is
        query_block x chunk x 4 B (2 GB by default) whatever the number of queries (the 503 k MS-MARCO train queries in one
        call would otherwise ask for 132 GB)."""
        assert dim % 4 == 0, "embedding size must be a multiple of 4"
        self.d, self.id_base, self.chunk, self.query_block = int(dim), int(id_base), int(chunk), int(query_block)
        self._parts, self._emb = [], None

    @property
    def ntotal(self):
        return sum(p.shape[0] for p in self._parts) + (0 if self._emb is None else self._emb.shape[0])

    def add(self, emb):
        emb = torch.as_tensor(emb)
        if not emb.is_cuda:
            raise L.SimxError("FlatIPIndex.add: embeddings must live on the GPU (there is no CPU path)")
        assert emb.dim() == 2 and emb.shape[1] == self.d
        self._parts.append(emb.detach().to(torch.float32).contiguous())

    def _corpus(self):
        if self._parts:
            self._emb = torch.cat(([self._emb] if self._emb is not None else []) + self._parts, 0)
            self._parts = []
        return self._emb

    def search(self, q, k, group=None):
        """-> (scores [nq,k] f32, ids [nq,k] int64), rows sorted by descending score (ties: ascending id); with
        ``group`` every rank passes the SAME queries and receives the global result."""
        q = torch.as_tensor(q)
        if not q.is_cuda:
            raise L.SimxError("FlatIPIndex.search: queries must live on the GPU (there is no CPU path)")
        q = q.detach().to(torch.float32).contiguous()
        nq = q.shape[0]
        corpus = self._corpus()
        nc = 0 if corpus is None else corpus.shape[0]
        D = torch.empty(nq, k, dtype=torch.float32, device=q.device)
        I = torch.empty(nq, k, dtype=torch.int64, device=q.device)
        chunk = max(4, min(self.chunk, max(nc, 4)))
        qb = max(1, min(self.query_block, nq))
        ws_bytes = int(L.load().simx_flat_ip_workspace_bytes(qb, chunk))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=q.device)
        for lo in range(0, nq, qb):                     # query blocks reuse ONE workspace
            n = min(qb, nq - lo)
            L.call("simx_flat_ip_search", L.stream_ptr(), n, nc, self.d, L.ptr(q[lo:lo + n]), L.ptr(corpus) if nc else None,
                   self.id_base, k, chunk, L.ptr(ws), ws_bytes, L.ptr(D[lo:lo + n]), L.ptr(I[lo:lo + n]))
        if group is None:
            return D, I
        return merge_topk(D, I, k, group)


def merge_topk(D, I, k, group):
    """all_gather the per-shard results (RCCL) and fold them with the top-k kernel; every rank gets the global top-k."""
    import torch.distributed as dist
    W = dist.get_world_size(group)
    Ds = [torch.empty_like(D) for _ in range(W)]
    Is = [torch.empty_like(I) for _ in range(W)]
    dist.all_gather(Ds, D, group=group)
    dist.all_gather(Is, I, group=group)
    return fold_candidates(torch.cat(Ds, 1).contiguous(), torch.cat(Is, 1).contiguous(), k)


def fold_candidates(cand_scores, cand_ids, k):
    """exact top-k of explicit candidates [nq,m] (ids int64, negative = padding)."""
    nq, m = cand_scores.shape
    D = torch.full((nq, k), float("-inf"), dtype=torch.float32, device=cand_scores.device)
    I = torch.full((nq, k), -1, dtype=torch.int64, device=cand_scores.device)
    L.call("simx_topk_update", L.stream_ptr(), nq, m, L.ptr(cand_scores), m, L.ptr(cand_ids), m, 0, k, L.ptr(D), L.ptr(I))
    return D, I
