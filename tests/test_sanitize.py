"""AddressSanitizer + UBSan over the library's HOST routines (-m "not gpu").

GPU AddressSanitizer is not available on the MI355X pool and the kernels have no CPU build; what runs under the sanitizers is
the part of libvbx_hip.so that never touches the device -- average linkage in both arithmetic forms, the flat-cluster cut, the
Kaldi archive index (vbhmm.py:117,140-146) -- compiled with g++ from the header the library itself includes
(vbx_amd/csrc/vbx_linkage.hpp; tests/sanitize/host_routines.cpp: random and adversarial inputs, buffers of exactly the stated
size, truncated and corrupted archives).  SURVEY.md section 5, row "sanitizers"."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_routines_are_clean_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which('g++')
    if not gxx:
        pytest.skip('g++ not available')
    exe = str(tmp_path / 'host_routines')
    build = subprocess.run([gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
                            '-I', os.path.join(REPO, 'vbx_amd', 'csrc'), '-o', exe,
                            os.path.join(REPO, 'tests', 'sanitize', 'host_routines.cpp')], capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and ('asan' in build.stderr.lower() or 'ubsan' in build.stderr.lower() or 'sanitize' in build.stderr.lower()):
        pytest.skip('this g++ has no sanitizer runtime: ' + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1'))
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    assert 'OK under the sanitizers' in run.stdout and 'ERROR: AddressSanitizer' not in run.stderr and 'runtime error' not in run.stderr
