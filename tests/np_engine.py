"""TEST INFRASTRUCTURE: a tiny CPU engine (numpy on torch CPU tensors) with the interface
pastix_amd.dist.factorize_levels expects, to exercise the multi-rank protocol over gloo."""
import numpy as np
import torch


class NumpyEngine:
    def __init__(self, cblk4, blok4, owner, level, rank, L0_full):
        self.c4 = np.asarray(cblk4, dtype=np.int64)
        self.b4 = np.asarray(blok4, dtype=np.int64)
        self.owner, self.level, self.rank = owner, level, rank
        nc = len(self.c4) - 1
        w = self.c4[:-1, 1] - self.c4[:-1, 0] + 1
        off = np.concatenate([[0], np.cumsum(w * self.c4[:-1, 3])])
        self.w = w
        self.panels_ = {}
        for k in range(nc):
            if owner[k] == rank:
                self.panels_[k] = torch.from_numpy(L0_full[off[k]:off[k + 1]].copy())
        for k in range(nc):
            if owner[k] != rank:
                continue
            for b in range(self.c4[k, 2] + 1, self.c4[k + 1, 2]):
                t = int(self.b4[b, 2])
                if t not in self.panels_:
                    self.panels_[t] = torch.zeros(int(w[t] * self.c4[t, 3]), dtype=torch.float64)   # shadow

    def panel(self, k):
        return self.panels_[k]

    def _mat(self, k):
        return self.panels_[k].numpy().reshape(int(self.w[k]), int(self.c4[k, 3])).T   # stride x width view

    def recv_numel(self, k, src):
        return int(self.panels_[k].numel())          # this test engine keeps full-size fan-in buffers

    def add(self, k, buf, src):
        self.panels_[k].add_(buf)

    # compact blocks of the native driver's schedule (pastix_amd.dist.factorize_scheduled)
    def pack(self, k, rows):
        """rows x width block of this rank's fan-in buffer for cblk k, column-major like the device's compact panels"""
        return torch.from_numpy(np.ascontiguousarray(self._mat(k)[rows, :].T).ravel().copy())

    def add_rows(self, k, rows, buf):
        w = int(self.w[k])
        self._mat(k)[rows, :] += buf.numpy().reshape(w, len(rows)).T

    def update(self, l):
        """contributions of every owned source of level l-1 (right-looking)"""
        c4, b4 = self.c4, self.b4
        for k in np.nonzero((self.level == l - 1) & (self.owner == self.rank))[0]:
            A = self._mat(k)
            fb, lb = c4[k, 2], c4[k + 1, 2]
            for i in range(fb + 1, lb):
                t = int(b4[i, 2])
                C = self._mat(t)
                Bi = A[b4[i, 3]:b4[i, 3] + b4[i, 1] - b4[i, 0] + 1]
                c0 = b4[i, 0] - c4[t, 0]
                b3 = c4[t, 2]
                for j in range(i, lb):
                    while not (b4[j, 0] >= b4[b3, 0] and b4[j, 1] <= b4[b3, 1]):
                        b3 += 1
                    Aj = A[b4[j, 3]:b4[j, 3] + b4[j, 1] - b4[j, 0] + 1]
                    r0 = b4[b3, 3] + b4[j, 0] - b4[b3, 0]
                    C[r0:r0 + Aj.shape[0], c0:c0 + Bi.shape[0]] -= Aj @ Bi.T

    def panels(self, l):
        for k in np.nonzero((self.level == l) & (self.owner == self.rank))[0]:
            A = self._mat(k)
            w = int(self.w[k])
            Ld = np.linalg.cholesky(np.tril(A[:w, :w]) + np.tril(A[:w, :w], -1).T)
            A[:w, :w] = np.tril(Ld) + np.triu(A[:w, :w], 1)
            if A.shape[0] > w:
                A[w:, :] = np.linalg.solve(Ld, A[w:, :].T).T
