"""numpy shard engine — TEST INFRASTRUCTURE ONLY.

A CPU stand-in with the same phase interface as rio-rs_amd/sharded.py::HipShardEngine, written
straight from the sharded-solve spec (include/rio_gpu_placement.h, "row-sharded solve").  It exists
so that the exchange protocol driven by `ShardedSolver` (what is all-gathered, in which order the
records are reduced, when the fix-up exchanges happen) can be checked against the whole-table CPU
oracle (oracle/placement_oracle.c orc_tick) on machines without a GPU, over gloo with
world_size 2.  It is never imported by the product package.
"""
import numpy as np

import spec_tick
import torch

NONE = 0xFFFFFFFF
SPILL = 0xFFFFFFFE
U64MAX = (1 << 64) - 1


def _u64(t):
    return t.numpy().view(np.uint64)


class CpuShardEngine:
    def __init__(self, cur, load, aff, cap, alive):
        self.cur = np.ascontiguousarray(cur, np.uint32)
        self.load = np.ascontiguousarray(load, np.uint32)
        self.aff = np.ascontiguousarray(aff, np.uint32)
        self.cap = np.ascontiguousarray(cap, np.uint64)
        self.alive = np.ascontiguousarray(alive, np.uint8).astype(bool)
        self.n, self.m = len(self.cur), len(self.cap)
        self.words1, self.words2 = 2 * self.m + 8, self.m + 2
        self.next = None
        self.assign = self.cur.copy()

    def new_buffer(self, words):
        return torch.zeros(int(words), dtype=torch.int64)

    # phase 1: classify rows, optimistic assignment, local per-node load sums
    def scan(self, x):
        m, c, a, l = self.m, self.assign, self.aff, self.load.astype(np.uint64)
        cin, ain = c < m, a < m
        kept = cin & self.alive[np.where(cin, c, 0)] if self.n else np.zeros(0, bool)
        cl = ~kept & ain & self.alive[np.where(ain, a, 0)] if self.n else np.zeros(0, bool)
        sp = ~kept & ~cl
        self.kept_mask, self.claim_mask = kept, cl
        self.next = np.where(kept, c, np.where(cl, a, SPILL)).astype(np.uint32)
        kl = np.zeros(m, np.uint64)
        np.add.at(kl, c[kept], l[kept])
        clm = np.zeros(m, np.uint64)
        np.add.at(clm, a[cl], l[cl])
        self.kept_local, self.claim_local = kl, clm
        self.st = dict(kept=int(kept.sum()), evicted=int((~kept & (c != NONE)).sum()), claimants=int(cl.sum()),
                       spillcand=int(sp.sum()), load_kept=int(kl.sum()), load_claim=int(clm.sum()),
                       rejected=0, load_rejected=0, spilled=0, load_spilled=0, unplaced=0, load_unplaced=0)
        X = _u64(x)
        X[:m] = kl
        X[m:2 * m] = clm
        X[2 * m:2 * m + 8] = [self.st["load_kept"], self.st["load_claim"], 0, self.st["kept"], self.st["evicted"],
                              self.st["claimants"], self.st["spillcand"], 1]

    # phase 2: global view from the gathered X records, reduced in rank order
    def resolve(self, rank, n_ranks, xg):
        m = self.m
        X = _u64(xg).reshape(n_ranks, self.words1)
        kept_glob = X[:, :m].sum(axis=0, dtype=np.uint64)
        claim_pre = X[:rank, m:2 * m].sum(axis=0, dtype=np.uint64) if rank else np.zeros(m, np.uint64)
        claim_glob = X[:, m:2 * m].sum(axis=0, dtype=np.uint64)
        fre = np.where(self.alive & (self.cap > kept_glob), self.cap - kept_glob, 0).astype(np.uint64)
        self.forced = claim_pre > fre
        self.free_local = np.where(self.forced, 0, fre - np.minimum(claim_pre, fre)).astype(np.uint64)
        self.gprev = kept_glob.copy()
        self.gfinal = kept_glob + claim_glob
        self.adm_local = self.claim_local.copy()
        self.rank, self.R = rank, n_ranks
        self.rank_base = 0
        cnt = X[:, 2 * m:2 * m + 8].sum(axis=0, dtype=np.uint64)
        self.info = dict(cut_nodes=int((claim_glob > fre).sum()), spill_rows=int(cnt[6]),
                         local_fixup=int((self.forced | (self.claim_local > self.free_local)).sum()),
                         kept=int(cnt[3]), evicted=int(cnt[4]), claimants=int(cnt[5]), load_kept=int(cnt[0]),
                         load_claim=int(cnt[1]), n_slow=0)
        self.info["n_slow"] = int(self.info["cut_nodes"] > 0 or self.info["spill_rows"] > 0)

    def verdict(self):
        self.slow = self.info["n_slow"] > 0
        if not self.slow:
            self.used = self.gfinal.copy()
        return dict(self.info)

    def _pending(self):
        return np.flatnonzero(self.next == SPILL)

    def _export(self, y, delta):
        Y = _u64(y)
        p = self._pending()
        Y[:self.m] = delta
        Y[self.m] = int(self.load[p].astype(np.uint64).sum())
        Y[self.m + 1] = len(p)

    # fix-up 1: strict index-ordered prefix cut of this rank's claimants against what is left for it
    def cut(self, run_local_fixup, y):
        if run_local_fixup:
            rows = np.flatnonzero(self.claim_mask)
            order = np.argsort(self.aff[rows], kind="stable")
            rows = rows[order]
            a = self.aff[rows]
            l = self.load[rows].astype(np.uint64)
            cs = np.cumsum(l, dtype=np.uint64)
            first = np.r_[True, a[1:] != a[:-1]] if len(a) else np.zeros(0, bool)
            start = np.maximum.accumulate(np.where(first, np.arange(len(a)), 0)) if len(a) else np.zeros(0, np.int64)
            base = np.where(start > 0, cs[np.maximum(start - 1, 0)], 0).astype(np.uint64) if len(a) else cs
            incl = cs - base                       # inclusive per-node prefix in index order
            over = (incl > self.free_local[a]) | self.forced[a]
            # strict cut: once a claimant overflows, everyone after it on that node is rejected too
            seen = np.zeros(len(a), bool)
            if len(a):
                grp = np.cumsum(first) - 1
                ov_i = np.where(over, np.arange(len(a)), len(a))
                first_over = np.full(grp.max() + 1, len(a))
                np.minimum.at(first_over, grp, ov_i)
                seen = np.arange(len(a)) >= first_over[grp]
            rej = rows[seen]
            self.next[rej] = SPILL
            self.st["rejected"] = len(rej)
            self.st["load_rejected"] = int(self.load[rej].astype(np.uint64).sum())
            adm = np.zeros(self.m, np.uint64)
            np.add.at(adm, a[~seen], l[~seen])
            self.adm_local = adm
        self._export(y, self.adm_local)

    def merge(self, yg):
        m = self.m
        Y = _u64(yg).reshape(self.R, self.words2)
        self.gprev = self.gprev + Y[:, :m].sum(axis=0, dtype=np.uint64)
        self.used = self.gprev.copy()
        self.rank_base = int(Y[:self.rank, m].sum(dtype=np.uint64)) if self.rank else 0
        return int(Y[:, m + 1].sum(dtype=np.uint64)), int(Y[:, m].sum(dtype=np.uint64))

    # fix-up 2: one water-fill round over this rank's pending rows
    def spill(self, rnd, last, y):
        fre = np.where(self.alive & (self.cap > self.used), self.cap - self.used, 0).astype(np.uint64)
        nz = np.flatnonzero(fre > 0)
        order = np.array(sorted(nz.tolist(), key=lambda j: (-spec_tick.capacity_class(int(fre[j])), j)), dtype=np.int64)
        Cs = [0]
        for j in order:
            Cs.append(min(Cs[-1] + int(fre[j]), U64MAX))
        Cv = np.array(Cs, dtype=np.uint64)
        p = self._pending()
        l = self.load[p].astype(np.uint64)
        adm = np.zeros(self.m, np.uint64)
        if len(p):
            Q = np.uint64(self.rank_base) + np.cumsum(l, dtype=np.uint64) - l
            node = np.full(len(p), NONE, np.uint32)
            if len(order):
                lo = np.searchsorted(Cv[:len(order) + 1], Q, side="right") - 1
                ok = Q < Cv[len(order)]
                lo_c = np.clip(lo, 0, len(order) - 1)
                fits = ok & (Q + l <= Cv[lo_c + 1])
                node[fits] = order[lo_c[fits]].astype(np.uint32)
            placed = node != NONE
            self.next[p[placed]] = node[placed]
            np.add.at(adm, node[placed], l[placed])
            self.st["spilled"] += int(placed.sum())
            self.st["load_spilled"] += int(l[placed].sum())
            if last:
                self.next[p[~placed]] = NONE
                self.st["unplaced"] += int((~placed).sum())
                self.st["load_unplaced"] += int(l[~placed].sum())
        self.used = self.used + adm
        self._export(y, adm)

    def finish(self):
        s = self.st
        return dict(n_objects=self.n, kept=s["kept"], evicted=s["evicted"], claimed=s["claimants"] - s["rejected"],
                    spilled=s["spilled"], unplaced=s["unplaced"], load_kept=s["load_kept"],
                    load_claimed=s["load_claim"] - s["load_rejected"], load_spilled=s["load_spilled"],
                    load_unplaced=s["load_unplaced"], cut_nodes=0, slow_path=0, rounds_run=0)

    def commit(self):
        self.assign = self.next.copy()

    def sync(self):
        pass
