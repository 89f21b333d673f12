"""GPU parity: HIP rollout buffer (store / GAE scan / get) vs the oracle and the reference's golden
vectors.  GAE outputs must be BIT-EXACT (float64 recurrence restated op for op); the standardised
advantages are compared at rtol 1e-5 (the reference reduces in float32 pairwise order, the kernels in
float64 -- SURVEY.md 8c)."""
import numpy as np
import pytest
import torch

import np_oracle as O

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _mk(T, N, D_o, D_a, est='gae', pc=0.0, lam_c=0.9, variant='sequential'):
    from omnisafe_amd.buffer import VectorOnPolicyBuffer
    from omnisafe_amd.spaces import Box

    return VectorOnPolicyBuffer(Box(-np.inf, np.inf, (D_o,)), Box(-1, 1, (D_a,)), size=T, gamma=0.99,
                                lam=0.95, lam_c=lam_c, advantage_estimator=est,
                                penalty_coefficient=pc, standardized_adv_r=True,
                                standardized_adv_c=True, num_envs=N, device=DEV, gae_variant=variant)


def _fill(buf, g_in, path_end, boot_r, boot_c, vector_finish=True):
    T, N = g_in['reward'].shape
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    for t in range(T):
        buf.store(**{k: dev(v[t]) for k, v in g_in.items()})
        if vector_finish:
            buf.finish_paths(dev(path_end[t]), dev(boot_r[t]), dev(boot_c[t]))
        else:
            for n in range(N):
                if path_end[t, n]:
                    buf.finish_path(torch.tensor([boot_r[t, n]]), torch.tensor([boot_c[t, n]]), n)


IN_KEYS = ('obs', 'act', 'reward', 'cost', 'value_r', 'value_c', 'logp')


@pytest.mark.parametrize('est', ['gae', 'gae-rtg', 'plain', 'vtrace'])
@pytest.mark.parametrize('pc', [0.0, 0.3])
@pytest.mark.parametrize('vector_finish', [True, False])
def test_golden_buffer(golden, est, pc, vector_finish):
    g = golden('buffer.npz')
    T, N = g['reward'].shape
    buf = _mk(T, N, g['obs'].shape[2], g['act'].shape[2], est, pc)
    _fill(buf, {k: g[k] for k in IN_KEYS}, g['path_end'], g['boot_r'], g['boot_c'], vector_finish)
    buf.compute_advantages()
    tag = f'{est}_pc{pc}'
    for k in ('adv_r', 'adv_c', 'target_value_r', 'target_value_c', 'discounted_ret'):
        got = O.env_major(buf.data[k].cpu().numpy())
        assert np.array_equal(got, g[f'{tag}/raw/{k}']), k  # bit-exact vs the reference
    data = buf.get()
    assert set(data) == {'obs', 'act', 'target_value_r', 'adv_r', 'logp', 'discounted_ret', 'adv_c',
                         'target_value_c'}
    for k in ('obs', 'act', 'logp', 'target_value_r', 'target_value_c', 'discounted_ret'):
        assert np.array_equal(data[k].cpu().numpy(), g[f'{tag}/get/{k}']), k
    np.testing.assert_allclose(data['adv_r'].cpu().numpy(), g[f'{tag}/get/adv_r'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(data['adv_c'].cpu().numpy(), g[f'{tag}/get/adv_c'], rtol=1e-5, atol=1e-6)
    assert buf.ptr == 0


def _random_case(rng, T, N, D_o=3, D_a=2, p_end=0.05):
    g_in = {
        'obs': rng.standard_normal((T, N, D_o)).astype(np.float32),
        'act': rng.standard_normal((T, N, D_a)).astype(np.float32),
        'reward': rng.standard_normal((T, N)).astype(np.float32),
        'cost': (rng.random((T, N)) < 0.05).astype(np.float32),
        'value_r': rng.standard_normal((T, N)).astype(np.float32),
        'value_c': rng.standard_normal((T, N)).astype(np.float32),
        'logp': rng.standard_normal((T, N)).astype(np.float32),
    }
    kind = rng.choice([0, 1, 2], size=(T, N), p=[1 - 2 * p_end, p_end, p_end])
    kind[-1] = 2
    boot_r = np.where(kind == 2, rng.standard_normal((T, N)), 0).astype(np.float32)
    boot_c = np.where(kind == 2, rng.standard_normal((T, N)), 0).astype(np.float32)
    return g_in, (kind != 0).astype(np.uint8), boot_r, boot_c


@pytest.mark.parametrize('T,N', [(16, 4096), (1, 64), (7, 1), (257, 300), (1000, 4), (64, 16384)])
def test_gae_vs_oracle_sizes(T, N):
    """Oracle (vectorised numpy restatement, pinned to the reference) at BASELINE sizes and ragged /
    degenerate shapes: single step, single env, T not a multiple of the scan chunk, N not a
    multiple of the wave."""
    rng = np.random.default_rng(T * 1000 + N)
    g_in, pe, br, bc = _random_case(rng, T, N)
    buf = _mk(T, N, 3, 2, lam_c=0.95)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    for k in IN_KEYS:
        buf.data[k].copy_(dev(g_in[k]))
    buf.data['path_end'].copy_(dev(pe))
    buf.data['boot_r'].copy_(dev(br))
    buf.data['boot_c'].copy_(dev(bc))
    buf.ptr = T
    ref = O.gae_time_major(g_in['reward'], g_in['cost'], g_in['value_r'], g_in['value_c'], pe, br, bc,
                           0.99, 0.95, 0.95)
    data = buf.get()
    for ok, bk in (('adv_r', 'adv_r'), ('adv_c', 'adv_c'), ('tgt_r', 'target_value_r'),
                   ('tgt_c', 'target_value_c'), ('disc_ret', 'discounted_ret')):
        assert np.array_equal(buf.data[bk].cpu().numpy(), ref[ok]), ok
    a_r, a_c, (mean_r, std_r, mean_c) = O.buffer_get(ref['adv_r'], ref['adv_c'])
    np.testing.assert_allclose(data['adv_r'].cpu().numpy(), a_r, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(data['adv_c'].cpu().numpy(), a_c, rtol=2e-5, atol=2e-6)
    st = buf.stats.cpu().numpy()
    assert st[2] == T * N
    np.testing.assert_allclose(st[4], mean_r, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(st[6], std_r, rtol=1e-5)
    assert np.array_equal(data['obs'].cpu().numpy(), O.env_major(g_in['obs']))
    assert np.array_equal(data['target_value_r'].cpu().numpy(), O.env_major(ref['tgt_r']))
    # size-independent properties of get(): zero-mean / unit-variance reward advantages,
    # zero-mean cost advantages
    x = data['adv_r'].double()
    assert abs(float(x.mean())) < 1e-4
    if T * N > 1:
        assert abs(float(x.pow(2).mean().sqrt()) - 1.0) < 1e-3
    assert abs(float(data['adv_c'].double().mean())) < 1e-4


def _load_case(buf, g_in, pe, br, bc):
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
    for k in ('reward', 'cost', 'value_r', 'value_c'):
        buf.data[k].copy_(dev(g_in[k]))
    buf.data['path_end'].copy_(dev(pe))
    buf.data['boot_r'].copy_(dev(br))
    buf.data['boot_c'].copy_(dev(bc))
    buf.ptr = buf.size


OUT_KEYS = ('adv_r', 'adv_c', 'target_value_r', 'target_value_c', 'discounted_ret')


@pytest.mark.parametrize('est', ['gae', 'gae-rtg', 'plain'])
@pytest.mark.parametrize('T,N,pc', [(16, 4096, 0.0), (1, 64, 0.0), (7, 1, 0.3), (64, 16, 0.0), (65, 33, 0.3),
                                    (257, 300, 0.0), (1000, 4, 0.3), (5000, 4, 0.0), (64, 16384, 0.0),
                                    (4096, 64, 0.0)])
def test_tiled_scan_equals_lane_per_env_kernel(est, T, N, pc):
    """osa_gae_scan_tiled (time-parallel wavefront scan, LDS-staged tiles of 64 steps x 16 envs) vs
    osa_gae_scan (bit-exact to the reference): same float32 deltas, float64 recurrences associated as a tree
    instead of a chain -> required rtol 1e-5 / atol 1e-6, and in practice bit-identical
    float32 outputs in > 99.9 % of the elements.  Shapes: tile-ragged T (1, 7, 65, 257, 1000, 5000 = BASELINE
    config 1), env blocks not a multiple of 16 (1, 4, 33, 300), carries across up to 79 tiles, paths ending
    on tile boundaries (random flags at 5 % per step)."""
    rng = np.random.default_rng(T * 7 + N)
    g_in, pe, br, bc = _random_case(rng, T, N, p_end=0.05 if T < 1000 else 0.002)
    outs = {}
    for variant in ('sequential', 'tiled'):
        buf = _mk(T, N, 3, 2, est, pc, lam_c=0.9, variant=variant)
        _load_case(buf, g_in, pe, br, bc)
        buf.compute_advantages()
        assert buf.last_gae_variant == variant
        outs[variant] = {k: buf.data[k].cpu().numpy().copy() for k in OUT_KEYS}
    same = total = 0
    for k in OUT_KEYS:
        a, b = outs['tiled'][k], outs['sequential'][k]
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6, err_msg=k)
        same += int((a == b).sum())
        total += a.size
    assert same / total > 0.999, same / total


@pytest.mark.parametrize('est', ['gae', 'gae-rtg', 'plain'])
@pytest.mark.parametrize('T,N,pc,p_end', [
    (16, 4096, 0.0, 0.05), (1, 64, 0.0, 0.05), (7, 1, 0.3, 0.05), (128, 64, 0.0, 0.05), (129, 33, 0.3, 0.05),
    (257, 300, 0.0, 0.05), (1000, 4, 0.3, 0.002), (5000, 70, 0.0, 0.0005), (256, 16384, 0.0, 0.01),
    (4096, 256, 0.0, 0.0), (2048, 512, 0.3, 0.001),
    # (added with the env-pair experiment of round 4, profiles/r4_gae_ab.md: a last workgroup with two live lanes, ...)
    (65, 130, 0.3, 0.05), (1000, 258, 0.3, 0.002), (4096, 128, 0.0, 0.0), (63, 1024, 0.0, 0.2)])
def test_chained_scan_equals_lane_per_env_kernel(est, T, N, pc, p_end):
    """osa_gae_scan_chained (time split over workgroups: levels of 128 steps, chunks of 16 steps in registers,
    per-env carries chained by a decoupled look-back) vs osa_gae_scan (bit-exact to the reference).  Given its
    incoming carry a chunk runs the sequential kernel's arithmetic step for step; the carry is assembled by the
    affine identity -> rtol 1e-5 / atol 1e-6 required, bit-identical float32 outputs in practice
    (> 99.99 %).  Shapes: level-ragged T (1, 7, 129, 257, 1000, 5000), env blocks not a multiple of 64 (1, 4, 33,
    70, 300), chains over up to 40 levels with NO path end at all (4096 x 256, p_end 0: the longest look-back) and
    with rare ones; run twice: the result must not depend on timing."""
    rng = np.random.default_rng(T * 11 + N)
    g_in, pe, br, bc = _random_case(rng, T, N, p_end=p_end)
    outs = {}
    for variant in ('sequential', 'chained', 'chained-again'):
        buf = _mk(T, N, 3, 2, est, pc, lam_c=0.9, variant=variant.split('-')[0])
        _load_case(buf, g_in, pe, br, bc)
        buf.compute_advantages()
        assert buf.last_gae_variant == variant.split('-')[0]
        outs[variant] = {k: buf.data[k].cpu().numpy().copy() for k in OUT_KEYS}
    same = total = 0
    for k in OUT_KEYS:
        a, b = outs['chained'][k], outs['sequential'][k]
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6, err_msg=k)
        assert np.array_equal(a, outs['chained-again'][k]), k  # deterministic: a pure function of the inputs
        same += int((a == b).sum())
        total += a.size
    assert same / total > 0.9999, same / total


def test_parallel_scans_propagate_nan_like_the_sequential_kernel():
    """A NaN reward (a diverged run) must neither hang the look-back -- carries are "not there yet" only while they hold
    the all-ones sentinel, any other NaN is a value -- nor spread differently from the sequential kernel: the same
    elements become NaN (the env's earlier steps up to its previous path end), everything else is unchanged."""
    rng = np.random.default_rng(3)
    T, N = 700, 70
    g_in, pe, br, bc = _random_case(rng, T, N, p_end=0.002)
    g_in['reward'][400, 5] = np.nan
    g_in['cost'][650, 69] = np.nan
    outs = {}
    for variant in ('sequential', 'chained', 'tiled'):
        buf = _mk(T, N, 3, 2, 'gae', 0.0, lam_c=0.9, variant=variant)
        _load_case(buf, g_in, pe, br, bc)
        buf.compute_advantages()
        torch.cuda.synchronize()
        outs[variant] = {k: buf.data[k].cpu().numpy().copy() for k in OUT_KEYS}
    for variant in ('chained', 'tiled'):
        for k in OUT_KEYS:
            a, b = outs[variant][k], outs['sequential'][k]
            assert np.array_equal(np.isnan(a), np.isnan(b)), (variant, k)
            np.testing.assert_allclose(a[~np.isnan(a)], b[~np.isnan(b)], rtol=1e-5, atol=1e-6, err_msg=k)
    assert np.isnan(outs['sequential']['adv_r'][:401, 5]).any() and not np.isnan(outs['sequential']['adv_r'][401:, 5]).any()


def test_chained_scan_vs_reference_golden(golden):
    """The chained kernel against the reference's own outputs (tests/golden/buffer.npz) for the three estimators it
    implements, with and without the cost penalty; v-trace falls back to the lane-per-env kernel."""
    g = golden('buffer.npz')
    T, N = g['reward'].shape
    for est in ('gae', 'gae-rtg', 'plain', 'vtrace'):
        for pc in (0.0, 0.3):
            buf = _mk(T, N, g['obs'].shape[2], g['act'].shape[2], est, pc, variant='chained')
            _fill(buf, {k: g[k] for k in IN_KEYS}, g['path_end'], g['boot_r'], g['boot_c'])
            buf.compute_advantages()
            assert buf.last_gae_variant == ('sequential' if est == 'vtrace' else 'chained')
            for k in OUT_KEYS:
                got = O.env_major(buf.data[k].cpu().numpy())
                np.testing.assert_allclose(got, g[f'{est}_pc{pc}/raw/{k}'], rtol=1e-5, atol=1e-6, err_msg=k)


def test_tiled_scan_vs_reference_golden(golden):
    """The tiled kernel against the reference's own outputs (tests/golden/buffer.npz: ragged paths, length-1
    paths, one path spanning the epoch) for the three estimators it implements, with and without the cost
    penalty; v-trace falls back to the lane-per-env kernel."""
    g = golden('buffer.npz')
    T, N = g['reward'].shape
    for est in ('gae', 'gae-rtg', 'plain', 'vtrace'):
        for pc in (0.0, 0.3):
            buf = _mk(T, N, g['obs'].shape[2], g['act'].shape[2], est, pc, variant='tiled')
            _fill(buf, {k: g[k] for k in IN_KEYS}, g['path_end'], g['boot_r'], g['boot_c'])
            buf.compute_advantages()
            assert buf.last_gae_variant == ('sequential' if est == 'vtrace' else 'tiled')
            for k in OUT_KEYS:
                got = O.env_major(buf.data[k].cpu().numpy())
                np.testing.assert_allclose(got, g[f'{est}_pc{pc}/raw/{k}'], rtol=1e-5, atol=1e-6, err_msg=k)


def test_gae_variant_auto_rule():
    from omnisafe_amd.buffer import VectorOnPolicyBuffer as B

    assert B.gae_variant_for(16, 4096, 0) == 'sequential'     # BASELINE config 2: a handful of steps per lane
    assert B.gae_variant_for(16, 1 << 20, 0) == 'sequential'
    assert B.gae_variant_for(32, 4096, 0) == 'sequential'
    assert B.gae_variant_for(5000, 4, 0) == 'chained'         # BASELINE config 1: 4 envs, long horizon
    assert B.gae_variant_for(4096, 4096, 0) == 'chained' and B.gae_variant_for(4096, 16, 0) == 'chained'
    assert B.gae_variant_for(256, 65536, 0) == 'chained' and B.gae_variant_for(64, 32768, 0) == 'chained'
    assert B.gae_variant_for(1024, 64, 0) == 'chained'
    assert B.gae_variant_for(64, 4096, 0) == 'tiled' and B.gae_variant_for(256, 256, 0) == 'tiled'
    assert B.gae_variant_for(256, 4096, 0) == 'tiled'
    assert B.gae_variant_for(5000, 4, 3) == 'sequential'      # v-trace
    buf = _mk(5000, 4, 3, 2, variant='auto')
    buf.ptr = 5000
    buf.compute_advantages()
    assert buf.last_gae_variant == 'chained'
    buf = _mk(128, 64, 3, 2, variant='auto')
    buf.ptr = 128
    buf.compute_advantages()
    assert buf.last_gae_variant == 'tiled'


def test_gae_linearity_full_size():
    """Size-independent property at the BASELINE size (N=4096, T=16): with values and bootstraps zero,
    GAE is linear in the reward stream: adv(a*r) = a*adv(r) for a power-of-two a (exact in fp)."""
    T, N = 16, 4096
    rng = np.random.default_rng(0)
    g_in, pe, br, bc = _random_case(rng, T, N)
    outs = []
    for scale in (1.0, 4.0):
        buf = _mk(T, N, 3, 2)
        buf.data['reward'].copy_(torch.from_numpy(g_in['reward'] * np.float32(scale)).to(DEV))
        buf.data['path_end'].copy_(torch.from_numpy(pe).to(DEV))
        buf.ptr = T
        buf.compute_advantages()
        outs.append(buf.data['adv_r'].cpu().numpy())
    assert np.array_equal(outs[0] * np.float32(4.0), outs[1])


def test_errors():
    from omnisafe_amd.buffer import VectorOnPolicyBuffer
    from omnisafe_amd.spaces import Box

    box = Box(-1, 1, (2,))
    with pytest.raises(ValueError):
        VectorOnPolicyBuffer(box, box, 4, .99, .95, .95, 'gae', 0.0, True, True, num_envs=0, device=DEV)
    with pytest.raises(NotImplementedError):
        VectorOnPolicyBuffer(object(), box, 4, .99, .95, .95, 'gae', 0.0, True, True, 1, DEV)
    with pytest.raises(NotImplementedError):
        VectorOnPolicyBuffer(box, box, 4, .99, .95, .95, 'bogus', 0.0, True, True, 1, DEV)
    buf = VectorOnPolicyBuffer(box, box, 1, .99, .95, .95, 'gae', 0.0, True, True, 1, DEV)
    z = torch.zeros(1, device=DEV)
    step = dict(obs=torch.zeros(1, 2, device=DEV), act=torch.zeros(1, 2, device=DEV), reward=z, cost=z,
                value_r=z, value_c=z, logp=z)
    buf.store(**step)
    with pytest.raises(AssertionError):
        buf.store(**step)


def test_cabi_error_codes():
    """Error behaviour of the boundary (include/omnisafe_amd.h): integer codes, no exceptions, nothing
    launched for bad arguments; the Python shim turns them into OsaError."""
    import ctypes as C

    from omnisafe_amd import _lib
    from omnisafe_amd.models import HParams, SurrogateExt

    lib = _lib.load(require_gpu=True)
    EINVAL, EUNSUPPORTED = -1, -3
    z = torch.zeros(64, 8, device=DEV)
    # null pointers / non-positive sizes
    assert lib.osa_gae_scan(None, None, None, None, None, None, None, 4, 4, 0.99, 0.95, 0.95, 0.0, 0,
                            None, None, None, None, None, None) == EINVAL
    assert lib.osa_vec_dot(0, _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), None) == EINVAL
    assert lib.osa_saute_step(4, None, None, None, None, None, None, 0.999, -1.0, None, None, None, 8, None, 0, 7,
                              None, None, None) == EINVAL
    # shapes the reference allows but the kernels do not cover: hidden widths other than 32 / 64 / 128 / 256, unknown
    # activation codes (bits 16-19 of the hidden word), act_dim > 32, obs too wide for the persistent kernel (which
    # also is 64-wide tanh only)
    out12 = (C.c_int * 12)()
    assert lib.osa_mlp_layout(60, 2, 96, out12) == EUNSUPPORTED
    assert lib.osa_mlp_layout(60, 2, 512, out12) == EUNSUPPORTED
    assert lib.osa_mlp_layout(60, 2, 64 | (7 << 16), out12) == EUNSUPPORTED
    assert lib.osa_mlp_layout(60, 2, 128 | (1 << 16), out12) == 0 and out12[10] == 128
    assert lib.osa_ppo_pass_supported(60, 2, 128) == 0 and lib.osa_ppo_pass_supported(60, 2, 64 | (1 << 16)) == 0
    assert lib.osa_mlp_layout(60, 40, 64, out12) == EUNSUPPORTED
    assert lib.osa_mlp_layout(60, 2, 64, out12) == 0 and out12[0] == 64 and out12[1] == 16
    assert lib.osa_ppo_pass_supported(60, 2, 64) == 1 and lib.osa_ppo_pass_supported(376, 17, 64) == 0
    # unaligned observation rows are refused by the persistent kernels (the host pads)
    hp = HParams()
    obs = torch.zeros(128, 27, device=DEV)
    v = torch.zeros(128, device=DEV)
    p = torch.zeros(3, 8448, device=DEV)
    st = torch.zeros(2, 16, device=DEV)
    step = torch.zeros(3, dtype=torch.int32, device=DEV)
    rc = lib.osa_ppo_pass(27, 2, 64, _lib.ptr(p), _lib.ptr(p), _lib.ptr(p), _lib.ptr(step), _lib.ptr(obs), 27,
                          _lib.ptr(obs), 27, _lib.ptr(v), _lib.ptr(v), _lib.ptr(v), _lib.ptr(v), _lib.ptr(v), None,
                          128, 64, _lib.ptr(v), C.byref(hp), 0, 7, _lib.ptr(st), None)
    assert rc == EUNSUPPORTED
    # extended surrogate: a mask or a penalty needs the whole minibatch in one block
    ext = SurrogateExt(cost_kappa=1.0)
    g = torch.zeros(3, 8448, device=DEV)
    rc = lib.osa_ppo_minibatch_ext(60, 2, 64, _lib.ptr(p), _lib.ptr(p), _lib.ptr(p), _lib.ptr(step), _lib.ptr(g),
                                   _lib.ptr(z), 60, _lib.ptr(z), 2, _lib.ptr(v), _lib.ptr(v), _lib.ptr(v),
                                   _lib.ptr(v), _lib.ptr(v), None, 128, _lib.ptr(v), C.byref(hp), 0, 0, 7, 4,
                                   _lib.ptr(g), _lib.ptr(st), C.byref(ext), None)
    assert rc == EUNSUPPORTED
    assert lib.osa_strerror(EINVAL) and lib.osa_strerror(EUNSUPPORTED)
    with pytest.raises(_lib.OsaError):
        _lib.check(EINVAL, 'osa_gae_scan')
    torch.cuda.synchronize()


@pytest.mark.parametrize('advantage_estimator', ['gae', 'vtrace', 'gae-rtg', 'plain'])
def test_vector_onpolicy_buffer_like_the_reference_test(advantage_estimator):
    """The reference's own test of this class (tests/test_buffer.py:29-168: obs/act dim 1, size 100,
    gamma = lam = lam_c = 0.9, two envs, the four estimators, (num_envs, 1)-shaped scalars) run against the
    drop-in: same constructor keywords, `buffers[idx].data[key][ptr - 1]` after every store, `path_start_idx
    == ptr` after finish_path, shapes of get() -- plus what the reference's test leaves out, the values."""
    from omnisafe_amd.buffer import VectorOnPolicyBuffer
    from omnisafe_amd.spaces import Box

    size, num_envs = 100, 2
    vector_buffer = VectorOnPolicyBuffer(
        obs_space=Box(low=-1, high=1, shape=(1,)), act_space=Box(low=-1, high=1, shape=(1,)), size=size,
        gamma=0.9, lam=0.9, advantage_estimator=advantage_estimator, standardized_adv_r=True,
        standardized_adv_c=True, lam_c=0.9, penalty_coefficient=0.0, device=torch.device(DEV), num_envs=num_envs)
    assert vector_buffer.num_buffers == num_envs
    assert vector_buffer.standardized_adv_c is True and vector_buffer.standardized_adv_r is True
    assert len(vector_buffer.buffers) == num_envs
    gen = torch.Generator(device='cpu').manual_seed(0)
    keys = ('obs', 'act', 'reward', 'cost', 'value_r', 'value_c', 'logp')
    stored = {k: [] for k in keys}
    for _ in range(size):
        step = {k: torch.rand((num_envs, 1), generator=gen).to(DEV) for k in keys}
        vector_buffer.store(**step)
        for k in keys:
            stored[k].append(step[k].cpu().numpy().reshape(num_envs))
        for idx, buffer in enumerate(vector_buffer.buffers):
            for k in keys:
                assert torch.allclose(buffer.data[k][buffer.ptr - 1].reshape(-1), step[k][idx]), k
    last = torch.randn(num_envs, 2, generator=gen)
    for idx, buffer in enumerate(vector_buffer.buffers):
        vector_buffer.finish_path(last[idx, :1].to(DEV), last[idx, 1:].to(DEV), idx)
        assert buffer.path_start_idx == buffer.ptr == size
    data = vector_buffer.get()
    assert data['obs'].shape == (size * num_envs, 1) and data['act'].shape == (size * num_envs, 1)
    assert set(data) == {'obs', 'act', 'logp', 'target_value_r', 'target_value_c', 'adv_r', 'adv_c',
                         'discounted_ret'}
    # values: the oracle's per-path arithmetic (pinned to the reference's OnPolicyBuffer) on the same numbers
    s = {k: np.stack(v) for k, v in stored.items()}  # (T, N)
    pe = np.zeros((size, num_envs), bool)
    pe[-1] = True
    br = np.zeros((size, num_envs), np.float32)
    bc = np.zeros((size, num_envs), np.float32)
    br[-1], bc[-1] = last[:, 0].numpy(), last[:, 1].numpy()
    ref = O.gae_time_major(s['reward'], s['cost'], s['value_r'], s['value_c'], pe, br, bc, 0.9, 0.9, 0.9,
                           estimator=advantage_estimator)
    assert np.array_equal(data['target_value_r'].cpu().numpy(), O.env_major(ref['tgt_r']))
    assert np.array_equal(data['target_value_c'].cpu().numpy(), O.env_major(ref['tgt_c']))
    a_r, a_c, _ = O.buffer_get(ref['adv_r'], ref['adv_c'])
    np.testing.assert_allclose(data['adv_r'].cpu().numpy(), a_r, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(data['adv_c'].cpu().numpy(), a_c, rtol=2e-5, atol=2e-6)
    assert np.array_equal(data['logp'].cpu().numpy(), O.env_major(s['logp']))
