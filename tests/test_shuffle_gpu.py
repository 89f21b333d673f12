"""osa_shuffle_rows (csrc/shuffle_kernels.hip): the minibatch shuffles of an update -- one random permutation per pass,
what the reference gets from DataLoader(shuffle=True) / torch.randperm (policy_gradient.py:357-377) -- as a keyed
bijection instead of a sort.  Index work: bit-exact against the numpy twin (oracle/np_oracle.py:shuffle_rows);
statistics: what a uniform shuffle must satisfy."""
import numpy as np
import pytest
import torch

import np_oracle as O

DEV = 'cuda:0'


def _shuffle(seeds, M):
    from omnisafe_amd import _lib

    lib = _lib.load(require_gpu=True)
    s = torch.as_tensor(np.asarray(seeds, dtype=np.int64), device=DEV)
    out = torch.empty((len(seeds), M), dtype=torch.int64, device=DEV)
    _lib.check(lib.osa_shuffle_rows(_lib.ptr(s), len(seeds), M, _lib.ptr(out), _lib.stream_ptr()), 'osa_shuffle_rows')
    return out.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize('M', [1, 2, 3, 5, 64, 100, 1000, 4096, 65536, 100003, 1 << 20])
def test_every_row_is_a_permutation_and_matches_the_numpy_twin_bit_for_bit(M):
    seeds = [0, 1, 12345678901234567, (1 << 62) - 1, 42]
    p = _shuffle(seeds, M)
    for r in p:
        assert np.array_equal(np.sort(r), np.arange(M))
    if M <= 100003:
        assert np.array_equal(p, O.shuffle_rows(seeds, M))
    if M > 8:
        assert not np.array_equal(p[0], p[1]) and not np.array_equal(p[0], np.arange(M))
    assert np.array_equal(p, _shuffle(seeds, M))  # a pure function of (seed, i)



@pytest.mark.gpu
def test_position_value_table_is_uniform():
    """Over 20 000 shuffles of 32 elements every (position, value) cell must be hit 625 +- 25 times: chi-square with
    31 x 31 degrees of freedom; and the same for a non-power-of-two size (cycle walking)."""
    for M in (32, 24):
        rows = 20000
        p = _shuffle(np.arange(rows, dtype=np.int64) * 2654435761 + 17, M)
        cnt = np.zeros((M, M))
        np.add.at(cnt, (np.tile(np.arange(M), rows), p.ravel()), 1)
        exp = rows / M
        chi = ((cnt - exp) ** 2 / exp).sum()
        dof = (M - 1) ** 2
        assert abs(chi - dof) < 5 * np.sqrt(2 * dof), (M, chi, dof)


@pytest.mark.gpu
def test_minibatch_composition_is_uniform_and_rows_are_independent():
    """What the update consumes: consecutive blocks of the permutation are the minibatches.  (a) neighbours are
    uncorrelated beyond the -1/(M-1) of sampling without replacement; (b) the number of elements of a fixed subset
    that land in the first minibatch is hypergeometric; (c) two passes' shuffles are uncorrelated."""
    M, rows, B = 4096, 512, 64
    p = _shuffle(np.arange(rows, dtype=np.int64) * 7919 + 3, M).astype(np.float64)
    c = np.corrcoef(p[:, :-1].ravel(), p[:, 1:].ravel())[0, 1]
    assert abs(c + 1.0 / (M - 1)) < 4.0 / np.sqrt(rows * (M - 1)), c
    low = (p[:, :B] < M / 4).sum(axis=1)  # hypergeometric: mean B / 4, variance B * 1/4 * 3/4 * (M - B) / (M - 1)
    var = B * 0.25 * 0.75 * (M - B) / (M - 1)
    assert abs(low.mean() - B / 4) < 4 * np.sqrt(var / rows), low.mean()
    assert abs(low.var(ddof=1) - var) < 0.25 * var, (low.var(ddof=1), var)
    c2 = np.corrcoef(p[0::2].ravel(), p[1::2].ravel())[0, 1]
    assert abs(c2) < 4.0 / np.sqrt(rows * M / 2), c2


@pytest.mark.gpu
def test_updater_shuffles_are_seeded_and_sort_switch_still_works(monkeypatch):
    import types

    from omnisafe_amd.models import ConstraintActorCritic
    from omnisafe_amd.spaces import Box
    from omnisafe_amd.update import PPOUpdater

    ns = types.SimpleNamespace
    mc = ns(actor=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
            critic=ns(hidden_sizes=[64, 64], activation='tanh', lr=3e-4),
            weight_initialization_mode='kaiming_uniform', actor_type='gaussian_learning', linear_lr_decay=True)
    ac = ConstraintActorCritic(Box(-np.inf, np.inf, (6,)), Box(-1, 1, (2,)), mc, 4, device=DEV)
    up = PPOUpdater(ac, batch_size=64, update_iters=3, target_kl=0.02, kl_early_stop=False)
    torch.manual_seed(11)
    a = up.shuffles(3, 1000).cpu().numpy()
    torch.manual_seed(11)
    b = up.shuffles(3, 1000).cpu().numpy()
    assert np.array_equal(a, b) and all(np.array_equal(np.sort(r), np.arange(1000)) for r in a)
    monkeypatch.setenv('OSA_SHUFFLE', 'sort')
    c = up.shuffles(3, 1000).cpu().numpy()
    assert all(np.array_equal(np.sort(r), np.arange(1000)) for r in c)
