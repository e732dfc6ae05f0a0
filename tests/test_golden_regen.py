"""The committed fixtures are what the committed generator produces.

Only where /root/reference exists (the build container): the small cases of tests/golden/make_golden.py are regenerated from the REAL reference into
a temp dir and compared with the committed files -- key sets equal, dtypes equal, integer / string arrays equal, floating-point arrays equal to CPU
thread-order noise.  "Pinned" holds only while script and data agree; this test is what notices when one of them moves without the other."""
import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')
SMALL = ['init', 'loss', 'bertadam', 'posbias', 'squeeze', 'fusion', 'augment3d', 'adversarial']       # seconds each on 8 threads

pytestmark = pytest.mark.skipif(not os.path.isdir('/root/reference/code'), reason='the reference is only present in the build container')


@pytest.fixture(scope='module')
def mg():
    sys.path.insert(0, GOLD)
    import make_golden
    return make_golden


@pytest.mark.parametrize('case', SMALL)
def test_committed_fixture_is_what_the_generator_writes(case, mg, tmp_path, monkeypatch):
    monkeypatch.setattr(mg, 'OUT_DIR', str(tmp_path))
    mg.CASES[case]()
    made = sorted(os.listdir(tmp_path))
    assert made, 'the case wrote nothing'
    for f in made:
        new, old = np.load(tmp_path / f, allow_pickle=False), np.load(os.path.join(GOLD, f), allow_pickle=False)
        assert sorted(new.files) == sorted(old.files), f
        for k in new.files:
            a, b = new[k], old[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (f, k, a.dtype, b.dtype, a.shape, b.shape)
            if a.dtype.kind in 'fc':
                scale = max(float(np.abs(b).max()) if b.size else 0.0, 1e-30)
                tol = 1e-12 if a.dtype == np.float64 else 1e-6          # fp32 results of multi-threaded CPU reductions: order noise only
                assert float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() if a.size else 0.0) <= tol * scale, (f, k)
            else:
                assert np.array_equal(a, b), (f, k)
