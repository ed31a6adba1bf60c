"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the LM-Net hot path.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / reported CPU baseline.  The product
package (``lm_net_amd``) never imports this package and fails loudly when its
HIP library is missing.
"""
