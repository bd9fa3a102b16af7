"""Plain CSR container used on the host side (int32 indices, float64 values)."""
import numpy as np


class CSR:
    """Row slab [row_begin, row_begin + nrows) of a matrix with `ncols` columns;
    column indices are GLOBAL.  Single rank: row_begin = 0."""

    def __init__(self, rowptr, colidx, val, ncols, row_begin=0):
        self.rowptr = np.ascontiguousarray(rowptr, np.int32)
        self.colidx = np.ascontiguousarray(colidx, np.int32)
        self.val = np.ascontiguousarray(val, np.float64)
        self.ncols = int(ncols)
        self.row_begin = int(row_begin)
        if self.rowptr.ndim != 1 or len(self.rowptr) < 1 or self.rowptr[0] != 0:
            raise ValueError("rowptr must be 1-D and start at 0")
        if len(self.colidx) != self.rowptr[-1] or len(self.val) != self.rowptr[-1]:
            raise ValueError("colidx/val length must equal rowptr[-1]")

    @property
    def nrows(self):
        return len(self.rowptr) - 1

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    def slab(self, r0, r1):
        """Rows [r0, r1) as a new slab (global columns kept)."""
        k0, k1 = int(self.rowptr[r0]), int(self.rowptr[r1])
        return CSR(self.rowptr[r0:r1 + 1] - k0, self.colidx[k0:k1], self.val[k0:k1], self.ncols,
                   self.row_begin + r0)

    @staticmethod
    def vstack(blocks):
        """Rows of several blocks with the same columns, one below the other (constraint blocks)."""
        rp = [np.zeros(1, np.int64)]
        off = 0
        for b in blocks:
            rp.append(b.rowptr[1:].astype(np.int64) + off)
            off += b.nnz
        return CSR(np.concatenate(rp).astype(np.int32), np.concatenate([b.colidx for b in blocks]),
                   np.concatenate([b.val for b in blocks]), blocks[0].ncols, 0)

    def col_slab(self, c0, c1):
        """Columns [c0, c1) of all rows (global columns kept) -- how the
        constraint block is dealt to ranks."""
        keep = (self.colidx >= c0) & (self.colidx < c1)
        counts = np.add.reduceat(keep.astype(np.int64), self.rowptr[:-1]) if self.nnz else np.zeros(self.nrows, np.int64)
        counts = np.where(np.diff(self.rowptr) == 0, 0, counts)
        rp = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        return CSR(rp, self.colidx[keep], self.val[keep], self.ncols, 0)
