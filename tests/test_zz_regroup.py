"""The development kernel that moves paths between walking and shading waves (csrc/paths_regroup.h) against the
default path kernel.  Collected last: it is not the path a render takes by default."""
import os
import subprocess
import sys

import pytest

from test_gpu_render import _TUNING_SNIPPET

pytestmark = pytest.mark.gpu


def test_regrouping_kernel_gives_the_same_images():
    """The development kernel that moves paths between walking and shading waves through rings in LDS
    (MIRT_PATHS_REGROUP) against the default path kernel: same bytes, same number of closest-hit queries."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outputs = []
    for walking in (None, 10, 5, 14):
        env = dict(os.environ)
        for k in ("MIRT_PATHS_REGROUP", "MIRT_REGROUP_EXTRA", "MIRT_REGROUP_TUNING", "MIRT_PATHS_TUNING"):
            env.pop(k, None)
        if walking is not None:
            env["MIRT_PATHS_REGROUP"] = str(walking)
        r = subprocess.run([sys.executable, "-c", _TUNING_SNIPPET, root], env=env, capture_output=True, text=True, timeout=110)
        assert r.returncode == 0, r.stderr[-2000:]
        outputs.append(r.stdout)
    for out in outputs[1:]:
        assert out == outputs[0]
