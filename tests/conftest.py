import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: longer statistical runs')


def pytest_sessionfinish(session, exitstatus):
    """The parity margins of this run (tests/util.py: every rel_err / allclose call site with the largest deviation it saw)."""
    try:
        import util
        util.write_margins(os.path.join(ROOT, 'gpurun_out', 'parity_margins.txt'))
    except Exception as e:                     # never turn a green run red over a report
        sys.stderr.write('parity margins not written: %r\n' % (e,))
