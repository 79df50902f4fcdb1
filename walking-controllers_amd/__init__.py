"""
walking-controllers_amd — MI355X-native batched QP solve path for the DCM-MPC and
Jacobian QP-IK of lia2790/walking-controllers.

  csrc/       HIP kernels + the C ABI (include/wcqp.h) + the C++ host mirror of the
              reference's WalkingController / WalkingQPIK interfaces
  capi.py     ctypes binding of the C ABI
  synth.py    synthetic iCub-shaped workloads
  sharding.py contiguous block split of the batch over ranks (+ optional scatter/gather)

The directory name carries a hyphen, so import it through the repo-root shim:
    import walking_controllers_amd as wca
"""
from . import capi, sharding, synth  # noqa: F401
from .capi import (IkSolver, MpcSolver, TickPipeline, KinModel, WcqpError, IK_FORM_OSQP, IK_FORM_QPOASES, IK_ALG_DEFAULT, IK_ALG_SWEEP, IK_ALG_NULLSPACE, IK_ALG_NULLSPACE_MFMA, IK_ALG_NULLSPACE_16L, IK_ALG_BASE_ELIM, IK_JAC_AUTO, IK_JAC_MIXED, IK_JAC_GENERAL, STATUS_STRUCTURE, KIN_HANDOFF_FUSED, KIN_HANDOFF_DENSE, KIN_HANDOFF_COMPACT,  # noqa: F401
                   STATUS_SOLVED, STATUS_MAX_ITER, STATUS_INFEASIBLE, STATUS_OUTSIDE_HULL,
                   STATUS_NUMERIC, device_count, hull_from_feet_host)
