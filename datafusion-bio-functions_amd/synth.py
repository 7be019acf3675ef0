"""Deterministic synthetic interval generator.

SplitMix64 counter-based stream, seed = 0x5EED0000 + 2*config + side (+ rank
salt for weak scaling); contig drawn proportional to hg38 contig length;
start ~ U[0, len-L], L = 1 + u mod (2*mean-1), closed end = start+L-1 (int32).
Rows are left unsorted.  The same arithmetic runs on torch (device) and numpy
(host) tensors, so the CPU baseline sees byte-identical inputs.
"""
import numpy as np

HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
        138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
        83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]

_G = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB


def _s64(x):
    """python int -> signed 64-bit two's complement"""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def splitmix64_np(seed, idx):
    """idx: uint64 array of counters -> uint64 outputs"""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (idx + np.uint64(1)) * np.uint64(_G))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(_M1)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(_M2)
        return z ^ (z >> np.uint64(31))


def _lsr(t, s):
    import torch
    return (t >> s) & ((1 << (64 - s)) - 1)


def splitmix64_torch(seed, idx):
    """idx: int64 tensor of counters -> int64 tensor holding the uint64 bit pattern"""
    z = (idx + 1) * _s64(_G) + _s64(seed)
    z = (z ^ _lsr(z, 30)) * _s64(_M1)
    z = (z ^ _lsr(z, 27)) * _s64(_M2)
    return z ^ _lsr(z, 31)


def gen_torch(n, mean, n_contigs, seed, device, chunk=1 << 24):
    """-> (key int32[n], start int32[n], end int32[n]) on `device`."""
    import torch
    lens = torch.tensor(HG38[:n_contigs], dtype=torch.int64, device=device)
    cum = torch.cumsum(lens, 0)
    total = int(cum[-1])
    key = torch.empty(n, dtype=torch.int32, device=device)
    start = torch.empty(n, dtype=torch.int32, device=device)
    end = torch.empty(n, dtype=torch.int32, device=device)
    m63 = (1 << 63) - 1
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        i = torch.arange(lo, hi, dtype=torch.int64, device=device)
        u0 = splitmix64_torch(seed, 3 * i) & m63          # 63-bit uniforms keep % non-negative
        u1 = splitmix64_torch(seed, 3 * i + 1) & m63
        u2 = splitmix64_torch(seed, 3 * i + 2) & m63
        k = torch.bucketize(u0 % total, cum, right=True)
        ln = 1 + u1 % (2 * mean - 1)
        st = u2 % (lens[k] - ln + 1)
        key[lo:hi] = k.to(torch.int32)
        start[lo:hi] = st.to(torch.int32)
        end[lo:hi] = (st + ln - 1).to(torch.int32)
    return key, start, end


def gen_numpy(n, mean, n_contigs, seed):
    lens = np.array(HG38[:n_contigs], dtype=np.uint64)
    cum = np.cumsum(lens)
    total = cum[-1]
    i = np.arange(n, dtype=np.uint64)
    m63 = np.uint64((1 << 63) - 1)
    u0 = splitmix64_np(seed, np.uint64(3) * i) & m63
    u1 = splitmix64_np(seed, np.uint64(3) * i + np.uint64(1)) & m63
    u2 = splitmix64_np(seed, np.uint64(3) * i + np.uint64(2)) & m63
    k = np.searchsorted(cum, u0 % total, side="right")
    ln = np.uint64(1) + u1 % np.uint64(2 * mean - 1)
    st = u2 % (lens[k] - ln + np.uint64(1))
    return k.astype(np.uint32), st.astype(np.int32), (st + ln - np.uint64(1)).astype(np.int32)


def key_counts_torch(n, n_contigs, seed, device, chunk=1 << 24):
    """rows per contig of gen_torch(n, ., n_contigs, seed) without materialising them (only the contig draw)"""
    import torch
    cum = torch.cumsum(torch.tensor(HG38[:n_contigs], dtype=torch.int64, device=device), 0)
    total = int(cum[-1])
    cnt = torch.zeros(n_contigs, dtype=torch.int64, device=device)
    m63 = (1 << 63) - 1
    for lo in range(0, n, chunk):
        i = torch.arange(lo, min(n, lo + chunk), dtype=torch.int64, device=device)
        k = torch.bucketize((splitmix64_torch(seed, 3 * i) & m63) % total, cum, right=True)
        cnt += torch.bincount(k, minlength=n_contigs)
    return cnt


def gen_torch_sharded(n, mean, n_contigs, seed, device, keep=None, lo=0, hi=None, chunk=1 << 24):
    """The rows of gen_torch(n, mean, n_contigs, seed) that one rank holds, generated chunk by chunk so that no rank
    ever materialises the whole job: rows lo..hi whose contig has keep[contig] set (keep: bool [n_contigs] or None).
    -> (key, start, end, rows) with rows = the kept rows' numbers in the whole job (int32)."""
    import torch
    hi = n if hi is None else hi
    lens = torch.tensor(HG38[:n_contigs], dtype=torch.int64, device=device)
    cum = torch.cumsum(lens, 0)
    total = int(cum[-1])
    m63 = (1 << 63) - 1
    parts = []
    for a in range(lo, hi, chunk):
        i = torch.arange(a, min(hi, a + chunk), dtype=torch.int64, device=device)
        k = torch.bucketize((splitmix64_torch(seed, 3 * i) & m63) % total, cum, right=True)
        if keep is not None:
            sel = keep[k]
            i, k = i[sel], k[sel]
        ln = 1 + (splitmix64_torch(seed, 3 * i + 1) & m63) % (2 * mean - 1)
        st = (splitmix64_torch(seed, 3 * i + 2) & m63) % (lens[k] - ln + 1)
        parts.append((k.to(torch.int32), st.to(torch.int32), (st + ln - 1).to(torch.int32), i.to(torch.int32)))
    if not parts:
        z = torch.zeros(0, dtype=torch.int32, device=device)
        return z, z.clone(), z.clone(), z.clone()
    return tuple(torch.cat([p[c] for p in parts]) for c in range(4))
