"""CPU oracle for the DeepCLR hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``deepclr_amd``) never does.

Pinning status (see DESIGN.md section "Oracle"):
  * composition (split/merge, set abstraction wiring, flow embedding, pose
    head, dual-quaternion -> 4x4): checked in the build container against the
    reference's own ``deepclr/models/{deepclr,helper,base}.py`` and
    ``deepclr/data/labels.py`` by ``tests/golden/make_golden.py``; the resulting
    vectors are committed under ``tests/golden/``.
  * third-party primitives (PointNet++ FPS / ball query / grouping,
    ``torch_cluster.knn``, transforms3d quaternion helpers): sources absent from
    the reference tree and no golden vector exists there -> PARITY UNPINNED;
    restated from the published algorithms in ``oracle/primitives.c`` and
    ``oracle/labels.py``.
"""
from .primitives import (ball_query, furthest_point_sample, gather_operation, grouping_operation, knn,
                         num_threads)
from .model import OracleDeepCLR, OracleSAModuleMSG, build_oracle_model
from .labels import dual_quat_to_matrix

__all__ = ['ball_query', 'furthest_point_sample', 'gather_operation', 'grouping_operation', 'knn',
           'num_threads', 'OracleDeepCLR', 'OracleSAModuleMSG', 'build_oracle_model',
           'dual_quat_to_matrix']
