"""CPU: the index arithmetic of the dead rows' limit-cycle shortcuts (k_mm_probe_head, k_mm_probe in csrc/tclip_kernels.hip), modelled
on a map with a known transient and period and compared with the brute-force trajectory for every checkpoint.

The kernels iterate a row whose states are s_0, s_1, ... (s_{l+1} = F(s_l)); the stop test needs, at every checkpoint iteration
l = 50, 100, ..., the pair measured on the step s_l -> s_{l+1}.  Once s_{j+1} equals an earlier snapshot s_b the trajectory is
periodic from b on with period p = j + 1 - b, and the pair of iteration l >= b is the one recorded for iteration b + (l - b) mod p.
This file restates exactly what the kernels do with their loop counters (head length l0, snapshots after 0, 8 and 32 probe
iterations, window of 64) and checks it for EVERY transient 0 .. 70, period 1 .. 40 and head length 1 .. 18."""
import pytest

K_MAX_CYCLE = 64


def make_map(mu, p):
    """states are integers: 0, 1, ..., mu + p - 1, then back to mu; pair(l) = a number that identifies the step taken at iteration l"""
    def state(l):
        return l if l < mu + p else mu + (l - mu) % p
    return state


def probe_head(state, l0, n_checks):
    """k_mm_probe_head: returns {m: iteration whose pair is written for checkpoint m} or None (row handed back)"""
    ref, ref8, have8, js, base, period = state(l0), None, False, 0, 0, 0
    cyc = {}
    j = 0
    while j < K_MAX_CYCLE and period == 0:
        cyc[j] = l0 + j                                   # pair of iteration l0 + j (the step s_{l0+j} -> s_{l0+j+1})
        cur = state(l0 + j + 1)
        if cur == ref:
            period, base = j + 1 - js, js
        elif have8 and cur == ref8:
            period, base = j + 1 - 8, 8
        elif j + 1 == 8:
            ref8, have8 = cur, True
        elif j + 1 == 32:
            ref, js = cur, 32
        j += 1
    if not period:
        return None
    out = {}
    for m in range(n_checks):
        t = 50 * (m + 1) - l0
        out[m] = cyc[base + (t - base) % period]
    return out


def probe_after_chunk(state, chunk, n_checks):
    """k_mm_probe after chunk `chunk` (iterations up to l1 = 50 (chunk + 1) have run; the pair of checkpoint `chunk` is cached)"""
    l1 = 50 * (chunk + 1)
    ref = state(l1 + 1)
    cyc, period = {}, 0
    for j in range(K_MAX_CYCLE):
        cyc[j] = l1 + 1 + j
        if state(l1 + 1 + j + 1) == ref:
            period = j + 1
            break
    if not period:
        return None
    return {m: cyc[(50 * (m + 1) - (l1 + 1)) % period] for m in range(chunk + 1, n_checks)}


@pytest.mark.parametrize("l0", [1, 4, 12, 16, 18])
def test_early_probe_fills_every_checkpoint_with_an_equivalent_iteration(l0):
    n_checks = 19
    found = handed_back = 0
    for mu in range(0, 71):
        for p in range(1, 41):
            state = make_map(mu, p)
            got = probe_head(state, l0, n_checks)
            if got is None:
                handed_back += 1
                # only rows whose cycle cannot be seen in the window: transient beyond the last snapshot or a period longer than what is left of it
                assert mu > l0 + 32 or p > K_MAX_CYCLE - 32 or (mu > l0 + 8 and 32 + p > K_MAX_CYCLE) or (mu > l0 and 8 + p > K_MAX_CYCLE), (mu, p)
                continue
            found += 1
            for m, it in got.items():
                l = 50 * (m + 1)
                # the iteration whose pair is used starts from the same state as iteration l does: same pair
                assert state(it) == state(l) and state(it + 1) == state(l + 1), (mu, p, m, it)
    assert found > 1000 and handed_back > 0
    # the rows the reference's trajectories produce (transient <= 28, period <= 20) are all found, after 12 + p, 20 + p or 44 + p iterations
    for mu in range(0, 29):
        for p in range(1, 21):
            assert probe_head(make_map(mu, p), 12, n_checks) is not None


@pytest.mark.parametrize("chunk", [0, 1])
def test_chunk_probe_indexing(chunk):
    n_checks = 19
    for mu in range(0, 120):
        for p in range(1, 70):
            state = make_map(mu, p)
            got = probe_after_chunk(state, chunk, n_checks)
            if got is None:
                assert mu > 50 * (chunk + 1) + 1 or p > K_MAX_CYCLE
                continue
            for m, it in got.items():
                l = 50 * (m + 1)
                assert state(it) == state(l) and state(it + 1) == state(l + 1), (mu, p, m)
