"""
The three parameter sets the reference ships (/root/reference/app/robots/<robot>/): VALUES only, as data - the numbers of
    controllerParams.ini            (Q, R, foot rectangle, hull tolerance, controllerHorizon 2 s -> N = 200)
    qpInverseKinematics.ini         (neck weight, additional_rotation, regularisation posture / weights / gains, task gains)
    zmpControllerParams.ini:7-8     (kZMP_walking, kCoM_walking)
    dcmWalkingCoordinator.ini:18    (com_height) and :2 / :9 (use_mpc, use_osqp)
so that every parity test can run on each of them, not only on iCubGazeboV2_5's.

    iCubGazeboV2_5   the set every round-1..3 test and golden used
    iCubGenova04     Q = 750, R = 9e7 (R/Q 100 x larger: the condensed KKT at N = 200 is the place to look), wider foot,
                     k_posFoot 7, k_attFoot 5; use_mpc and use_osqp commented out
    icubGazeboSim    the ONLY robot that ships `use_mpc 1`: com_height 0.49, another additional_rotation, every
                     regularisation weight and gain 0.5, k_posCom 1.5, k_posFoot 2.5, k_attFoot 5, k_neck 0.5
"""
import numpy as np

_POSTURE = [15, 0, 0, -7, 22, 11, 30, -7, 22, 11, 30, 5.082, 0.406, -0.131, -45.249, -26.454, -0.351, 5.082, 0.406, -0.131, -45.249, -26.454, -0.351]
_W_V25 = [1.0] * 3 + [2.0] * 8 + [1.0] * 12

ROBOTS = {
    "iCubGazeboV2_5": dict(
        Q=7500.0, R=9.0e6, foot_size=((-0.02, 0.05), (-0.025, 0.025)), hull_tol=0.05, com_height=0.53, k_zmp=3.0, k_com=9.0,
        neck_weight=5.0, additional_rotation=((0.0, 0.0, 1.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)),
        reg_deg=_POSTURE, reg_w=_W_V25, reg_k=[5.0] * 23, k_pos_com=1.0, k_pos_foot=4.0, k_att_foot=2.0, k_neck=1.0,
        use_mpc=False, use_osqp=True),
    "iCubGenova04": dict(
        Q=750.0, R=9.0e7, foot_size=((-0.02, 0.05), (-0.045, 0.05)), hull_tol=0.05, com_height=0.53, k_zmp=3.5, k_com=10.0,
        neck_weight=5.0, additional_rotation=((0.0, 0.0, 1.0), (1.0, 0.0, 0.0), (0.0, 1.0, 0.0)),
        reg_deg=_POSTURE, reg_w=_W_V25, reg_k=[5.0] * 23, k_pos_com=1.0, k_pos_foot=7.0, k_att_foot=5.0, k_neck=1.0,
        use_mpc=False, use_osqp=False),
    "icubGazeboSim": dict(
        Q=7500.0, R=9.0e6, foot_size=((-0.02, 0.05), (-0.025, 0.025)), hull_tol=0.05, com_height=0.49, k_zmp=1.7, k_com=5.5,
        neck_weight=5.0, additional_rotation=((0.0, -1.0, 0.0), (1.0, 0.0, 0.0), (0.0, 0.0, 1.0)),
        reg_deg=_POSTURE, reg_w=[0.5] * 23, reg_k=[0.5] * 23, k_pos_com=1.5, k_pos_foot=2.5, k_att_foot=5.0, k_neck=0.5,
        use_mpc=True, use_osqp=True),
}
NAMES = list(ROBOTS)


def mpc_params(qs, robot, horizon=50):
    r = ROBOTS[robot]
    return qs.MPCParams(horizon=horizon, com_height=r["com_height"], Q=r["Q"] * np.eye(2), R=r["R"] * np.eye(2),
                        convex_hull_tolerance=r["hull_tol"], foot_size=r["foot_size"])


def mpc_solver(wca, robot, horizon=50):
    r = ROBOTS[robot]
    return wca.MpcSolver(horizon=horizon, com_height=r["com_height"], Q=r["Q"] * np.eye(2), R=r["R"] * np.eye(2),
                         convex_hull_tolerance=r["hull_tol"])


def ik_params(qs, robot, v_max):
    r = ROBOTS[robot]
    return qs.IKParams(neck_weight=r["neck_weight"] * np.eye(3), additional_rotation=np.array(r["additional_rotation"]),
                       joint_reg_deg=np.array(r["reg_deg"], float), joint_reg_weights=np.array(r["reg_w"], float),
                       joint_reg_gains=np.array(r["reg_k"], float), k_pos_com=r["k_pos_com"], k_pos_foot=r["k_pos_foot"],
                       k_att_foot=r["k_att_foot"], k_neck=r["k_neck"], v_max=np.broadcast_to(np.asarray(v_max, float), (23,)).copy())


def ik_solver(wca, robot, form, v_max, **kw):
    r = ROBOTS[robot]
    return wca.IkSolver(form=wca.IK_FORM_QPOASES if form == "qpoases" else wca.IK_FORM_OSQP, neck_weight=r["neck_weight"] * np.eye(3),
                        joint_reg_weights=np.array(r["reg_w"], float), joint_reg_gains=np.array(r["reg_k"], float),
                        joint_reg_rad=np.deg2rad(np.array(r["reg_deg"], float)), v_max=v_max, k_pos_com=r["k_pos_com"],
                        k_pos_foot=r["k_pos_foot"], k_att_foot=r["k_att_foot"], k_neck=r["k_neck"], **kw)


def _tuple(v):
    return "(" + ", ".join("%.10g" % x for x in v) + ")"


def mpc_ini(robot, controller_horizon=0.5):
    """controllerParams.ini (+ the GENERAL group's sampling_time / com_height) of a robot in the reference's syntax;
    controllerHorizon 0.5 s = BASELINE's N = 50, 2 = the shipped N = 200."""
    r = ROBOTS[robot]
    (x0, x1), (y0, y1) = r["foot_size"]
    return ("controllerHorizon       %g\nsampling_time           0.01\ncom_height              %g\n\n"
            "stateWeightTriplets     ((0,0,%.10g), (1,1,%.10g))\ninputWeightTriplets     ((0,0,%.10g), (1,1,%.10g))\n\n"
            "foot_size               ((%g   %g), (%g   %g))\ninitial_zmp_position    (0.0 0.0)\n\nconvex_hull_tolerance   %g\n"
            % (controller_horizon, r["com_height"], r["Q"], r["Q"], r["R"], r["R"], x0, x1, y0, y1, r["hull_tol"]))


def ik_ini(robot):
    """qpInverseKinematics.ini of a robot in the reference's syntax."""
    r = ROBOTS[robot]
    rot = ",".join("(" + " ".join("%.1f" % x for x in row) + ")" for row in r["additional_rotation"])
    return ("useCoMAsConstraint               1\nneckWeightTriplets              ((0,0,%g), (1,1,%g), (2,2,%g))\n"
            "additional_rotation             (%s)\njointRegularization            %s\njointRegularizationWeights     %s\n"
            "jointRegularizationGains       %s\nk_posCom                        %g\nk_posFoot                       %g\n"
            "k_attFoot                       %g\nk_neck                          %g\n"
            % (r["neck_weight"], r["neck_weight"], r["neck_weight"], rot, _tuple(r["reg_deg"]), _tuple(r["reg_w"]), _tuple(r["reg_k"]),
               r["k_pos_com"], r["k_pos_foot"], r["k_att_foot"], r["k_neck"]))
