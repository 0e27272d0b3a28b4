"""CPU oracle for the MixQ W8A8O16 linear path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product package ``mixq_tensorrt_llm_amd`` never does (tests/test_layout.py checks that).
"""
from .oracle import *  # noqa: F401,F403
