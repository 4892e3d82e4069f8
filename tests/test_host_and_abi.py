"""CPU-only checks: the C-ABI library builds, loads and exports every declared symbol; the
host-side mirror reproduces the reference's argument handling; the product path refuses to
run without a GPU instead of silently falling back to anything."""
import os
import re
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, 'include', 'vbx_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vbx_[a-z_0-9]+)\s*\(', text)))


def test_library_builds_and_exports_the_declared_abi():
    from vbx_amd import _capi, build
    path = build.build()
    assert os.path.exists(path)
    lib = _capi.load()
    assert lib.vbx_abi_version() == 7
    syms = declared_symbols()
    assert set(syms) == set(_capi.ABI_SYMBOLS), set(syms) ^ set(_capi.ABI_SYMBOLS)
    exported = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True).stdout
    for s in syms:
        assert re.search(rf'\bT {s}\b', exported), f'{s} is declared in include/vbx_hip.h but not exported'


def test_code_object_is_gfx950_and_uses_mfma_and_dpp():
    from vbx_amd import build
    path = build.build()
    blob = open(path, 'rb').read()
    assert b'gfx950' in blob
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        pytest.skip('llvm-objdump not available')
    # the device code object sits in the .hip_fatbin section of the shared object
    import tempfile
    llvm = os.path.dirname(objdump)
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run([f'{llvm}/llvm-objcopy', '--dump-section', f'.hip_fatbin={tmp}/fat.bin', path],
                       capture_output=True)
        subprocess.run([f'{llvm}/clang-offload-bundler', '--unbundle', '--type=o',
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--input={tmp}/fat.bin',
                        f'--output={tmp}/dev.co'], capture_output=True)
        if not os.path.exists(f'{tmp}/dev.co'):
            pytest.skip('could not unbundle the device code object')
        dis = subprocess.run([objdump, '-d', f'{tmp}/dev.co'], capture_output=True, text=True).stdout
    assert 'v_mfma_f32_16x16x4_f32' in dis or 'v_mfma_f32_16x16x4f32' in dis
    assert 'v_mfma_f64_16x16x4_f64' in dis or 'v_mfma_f64_16x16x4f64' in dis
    assert 'row_mirror' in dis and 'quad_perm' in dis


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    import vbx_amd
    from vbx_amd import _capi
    X = np.random.default_rng(0).standard_normal((10, 16))
    with pytest.raises(_capi.VbxError):
        vbx_amd.VBx(X, np.ones(16), pi=3, gamma=np.full((10, 3), 1 / 3), maxIters=2)
    # the batch call and the pinned result blocks of ABI 7 alike: no device, no silent stand-in
    from vbx_amd.batch import VBx_batch
    with pytest.raises(_capi.VbxError):
        VBx_batch([dict(X=X, Phi=np.ones(16), pi=3, gamma=np.full((10, 3), 1 / 3))], maxIters=2)
    with pytest.raises(_capi.VbxError):
        _capi.pinned_arrays([(4, 4)])


def test_product_code_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, 'vbx_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h')):
                src = open(os.path.join(root, f)).read()
                assert 'oracle' not in src.replace('the oracle here', '').replace('oracle/chunked_scan.py', ''), f
    for f in ('VBx.py', 'diarization_lib.py'):
        assert 'oracle' not in open(os.path.join(REPO, 'vbx_drop_in', f)).read()


def test_argument_handling_before_the_device_is_touched():
    import vbx_amd
    X = np.random.default_rng(0).standard_normal((20, 16))
    Phi = np.ones(16)
    with pytest.raises(AssertionError):                       # VBx.py:85
        vbx_amd.VBx(X, Phi, pi=4, gamma=np.full((20, 5), 0.2))
    with pytest.raises(TypeError):                            # VBx.py:76: numpy ints are not `int`
        vbx_amd.VBx(X, Phi, pi=np.int64(4))
    g0 = np.full((20, 4), 0.25)
    gamma, pi, Li = vbx_amd.VBx(X, Phi, pi=4, gamma=g0, maxIters=0)
    assert gamma is g0 and Li == [] and np.allclose(pi, 0.25)
    out = vbx_amd.VBx(X, Phi, pi=4, gamma=g0, maxIters=0, return_model=True)
    assert len(out) == 5 and out[3] is None and out[4] is None


def test_der_matches_oracle():
    from oracle import vbx_oracle
    import vbx_amd
    rng = np.random.default_rng(5)
    q = rng.random((200, 6))
    q /= q.sum(1, keepdims=True)
    ref = rng.integers(0, 4, 200)
    for kw in ({}, {'expected': False}, {'xentropy': True}, {'expected': False, 'xentropy': True}):
        assert np.isclose(vbx_amd.DER(q, ref, **kw), vbx_oracle.DER(q, ref, **kw), rtol=1e-12), kw


def test_transition_structure_detection():
    from vbx_amd.VBx import _split_transition
    pi = np.array([0.2, 0.5, 0.3])
    lp, off = _split_transition(np.eye(3) * 0.9 + 0.1 * pi)
    assert np.isclose(lp, 0.9) and np.allclose(off, 0.1 * pi)
    # anything else goes to the dense kernel (vbx_forward_backward_dense): no structure to exploit
    assert _split_transition(np.array([[0.5, 0.5, 0.0], [0.1, 0.8, 0.1], [0.3, 0.3, 0.4]])) is None
    assert _split_transition(np.diag([0.9, 0.8, 0.7]) + 0.1 * pi) is None          # self-loop probability not constant
    assert _split_transition(np.eye(3) * 1.5 - 0.5 * pi) is None                   # loopProb outside [0, 1]


def test_batch_api_rejects_arguments_it_would_ignore():
    from vbx_amd.batch import _normalise
    rec = dict(X=np.zeros((4, 3)), Phi=np.ones(3), pi=2, gamma=np.full((4, 2), 0.5))
    assert _normalise(rec, dict(Fa=0.3))['Fa'] == 0.3
    with pytest.warns(UserWarning, match='maxIters'):             # a dict built for VBx(**kw): the per-batch arguments
        assert _normalise(dict(rec, maxIters=3, return_model=True, ref=None), {})['Fb'] == 1.0     # are ignored, loudly
    with pytest.raises(TypeError, match='looprob'):                 # a typo is an error, not silence
        _normalise(rec, dict(looprob=0.5))
    with pytest.raises(TypeError, match='Fc'):
        _normalise(dict(rec, Fc=1.0), {})


def test_drop_in_module_exports_reference_names():
    import importlib.util
    spec = importlib.util.spec_from_file_location('VBx_dropin', os.path.join(REPO, 'vbx_drop_in', 'VBx.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import inspect
    sig = inspect.signature(mod.VBx)
    names = list(sig.parameters)
    assert names[:15] == ['X', 'Phi', 'loopProb', 'Fa', 'Fb', 'pi', 'gamma', 'maxIters', 'epsilon', 'alphaQInit',
                          'ref', 'plot', 'return_model', 'alpha', 'invL']          # VBx.py:27-29
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d['loopProb'], d['Fa'], d['Fb'], d['pi'], d['maxIters'], d['epsilon'], d['alphaQInit']) == \
        (0.9, 1.0, 1.0, 10, 10, 1e-4, 1.0)
    assert callable(mod.forward_backward) and callable(mod.DER)


def test_sweep_api_host_side():
    """VBx_sweep: argument handling before the device is touched (the GPU tests compare its results with VBx() calls)."""
    from vbx_amd.batch import VBx_sweep
    X = np.random.default_rng(0).standard_normal((12, 8))
    Phi = np.ones(8)
    g0 = np.full((12, 3), 1 / 3)
    out = VBx_sweep(X, Phi, [dict(Fa=0.2), dict(Fa=0.4, Fb=6.0)], maxIters=0, pi=3, gamma=g0, return_model=True)
    assert len(out) == 2 and all(len(t) == 5 and t[0] is g0 and t[2] == [] and t[3] is None for t in out)
    assert VBx_sweep(X, Phi, [], maxIters=5) == []
    with pytest.raises(TypeError, match='Fc'):
        VBx_sweep(X, Phi, [dict(Fc=1.0)], maxIters=0, pi=3, gamma=g0)
    with pytest.raises(AssertionError):                           # VBx.py:85 per point
        VBx_sweep(X, Phi, [dict(pi=4)], maxIters=0, gamma=g0)


def test_bench_last_line_is_compact_and_complete():
    """bench.py: the LAST stdout line (what the driver parses) is a compact JSON < 4 KB with the contract's keys, `roofline`
    and `cpu_baseline` -- formatted here from a canned full record (the closing bench line of round 4, 25 KB, which the
    driver could not parse) with today's key names patched in."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(REPO, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    full = json.load(open(os.path.join(REPO, 'profiles', 'r04_bench_driver_args.json')))
    assert len(json.dumps(full)) > 20000                      # (the record that went unparsed)
    for roof in [full['roofline']] + [full[k]['roofline'] for k in ('f32_split', 'f64')]:
        roof['bound_today'], roof['bound'] = roof['bound'], 'hbm'
    full['cpu_baseline']['kind'] = 'port'
    line = bench.compact_record(full)
    assert '\n' not in line and len(line) < 4096, len(line)
    c = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'configs'):
        assert k in c, k
    assert c['steps'] == 20 and c['warmup'] == 5 and c['n_gpus'] == 1 and 'workload' in c['config']
    assert abs(c['value'] - full['value']) < 1e-6 * full['value']
    assert abs(c['ms_per_step'] - full['ms_per_step']) < 1e-6 * full['ms_per_step']
    assert set(c['roofline']) >= {'bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source',
                                  'algorithmic_bytes_per_launch', 'avg_launch_us'}
    assert c['roofline']['bound'] in ('hbm', 'mfma') and abs(c['roofline']['frac'] - c['roofline']['achieved'] / c['roofline']['peak']) < 1e-4
    assert set(c['cpu_baseline']) == {'value', 'unit', 'cores', 'kind', 'sample'} and c['cpu_baseline']['kind'] == 'port'
    assert set(full['configs']) <= set(c['configs']) and 'f32_split' in c['configs'] and 'f64' in c['configs']
    for name, d in c['configs'].items():
        assert set(d) <= {'value', 'ms', 'frac', 'bound_today'} and d['value'] is not None, name
    # a record three times the size still fits: entries are dropped from the end and the line says so
    big = dict(full, configs={f'{k}_{i}': v for k, v in full['configs'].items() for i in range(6)})
    line = bench.compact_record(big)
    assert len(line) < 4096 and json.loads(line).get('configs_truncated') is True
