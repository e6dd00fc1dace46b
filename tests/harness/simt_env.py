"""TKAMD_SIMT=1: make THIS process run the -m gpu tests against the host build of the kernels (tests/harness/simt_build.py).

install() is called by tests/conftest.py for the pytest process and by tests/harness/simt_site/sitecustomize.py for every python
subprocess a test starts (the A/B-variant tests select kernels through environment variables read once per process).  It
  * opens the SIMT build where tokenizers_amd._lib would open libtokenizers_amd.so (the product knows nothing about it), and
  * shrinks the synthetic corpora of oracle/synth.py (n / 200 lines, at least 300): the emulation runs one workgroup at a time.
Test infrastructure only."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
_installed = False


def scale(n: int, floor: int = 300, div: int = 200) -> int:
    return min(n, max(floor, n // div))


def install() -> None:
    global _installed
    if _installed:
        return
    _installed = True
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from tests.harness import simt_build
    from tokenizers_amd import _lib
    from oracle import synth
    _lib.LIB_PATH, _lib._lib = simt_build.build(), None
    gen_lines, stress_lines, zipf = synth.gen_lines, synth.stress_lines, synth.zipf_length_docs

    def gen_lines_scaled(n_lines, *a, **k):
        return gen_lines(scale(n_lines), *a, **k)

    def stress_lines_scaled(seed=0, n=2000):
        return stress_lines(seed=seed, n=scale(n, floor=150, div=20))

    def zipf_scaled(total_bytes, *a, **k):
        return zipf(scale(total_bytes, floor=200_000, div=200), *a, **k)

    synth.gen_lines, synth.stress_lines, synth.zipf_length_docs = gen_lines_scaled, stress_lines_scaled, zipf_scaled
    site = os.path.join(HERE, "simt_site")
    pp = os.environ.get("PYTHONPATH", "")
    if site not in pp.split(os.pathsep):
        os.environ["PYTHONPATH"] = site + (os.pathsep + pp if pp else "")
