import numpy as np
rng = np.random.default_rng(1)
n = 4_999_985
def pack_of(bitmap, p):
    # hash4j: prefix = bitmap << (p-1); register = 4*top + two bits below
    x = bitmap.astype(object)
    return None
for p in (18, 20, 22):
    m = 1 << p
    h = rng.integers(0, 2**63, size=n, dtype=np.int64).astype(np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n).astype(np.uint64)
    idx = (h >> np.uint64(64 - p)).astype(np.int64)
    low = (h << np.uint64(p)) | np.uint64((1 << p) - 1)
    # nlz = leading zeros of low (64-bit)
    nlz = np.zeros(n, np.int64)
    x = low.copy()
    for sh in (32, 16, 8, 4, 2, 1):
        mask = x < (np.uint64(1) << np.uint64(64 - sh))
        nlz += np.where(mask, sh, 0)
        x = np.where(mask, x << np.uint64(sh), x)
    order = np.argsort(idx, kind="stable")
    idx_s, nlz_s = idx[order], nlz[order]
    start = np.searchsorted(idx_s, np.arange(m))
    cnt = np.diff(np.append(start, n))
    # state: top (max nlz seen, -1 none), b1 (top-1 seen), b2 (top-2 seen)
    top = np.full(m, -1, np.int64); b1 = np.zeros(m, bool); b2 = np.zeros(m, bool)
    changes = 0; above_or_eq = 0
    for r in range(cnt.max()):
        sel = np.nonzero(cnt > r)[0]
        v = nlz_s[start[sel] + r]
        t = top[sel]; c1 = b1[sel]; c2 = b2[sel]
        d = t - v
        newtop = v > t
        set1 = (d == 1) & ~c1
        set2 = (d == 2) & ~c2
        ch = newtop | set1 | set2
        changes += int(ch.sum())
        # update
        nt = np.where(newtop, v, t)
        shift = np.where(newtop & (t >= 0), v - t, 0)
        n1 = np.where(newtop, (shift == 1), c1 | set1)          # old top becomes bit below when shift == 1
        n2 = np.where(newtop, ((shift == 2) | ((shift == 1) & c1)), c2 | set2)
        top[sel] = nt; b1[sel] = n1; b2[sel] = n2
    print("p=%d: %.2f k-mers per register; entries that change their register given PERFECT knowledge of it: %.1f %% of the k-mers; registers touched %.1f %%"
          % (p, n / m, 100.0 * changes / n, 100.0 * (cnt > 0).mean()))
    # the same per work item of a third of the genome (tables start empty per item): approx by n/3
