"""CPU oracle for the KBDM ensemble hot path -- TEST INFRASTRUCTURE ONLY.

Nothing in the shipped package (``llckbdm_amd``) may import this package.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
use it, and only as the checker / the timed CPU baseline.
"""
