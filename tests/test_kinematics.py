"""batched kinematics.  CPU: the oracle's analytic Jacobians against central differences of its own
forward kinematics (the reference's model and iDynTree are absent: parity unpinned).  GPU: the kernel against the
oracle, and the kernel's Jacobians driving the IK kernel against the exact IK optimum on the oracle's Jacobians."""
import numpy as np
import pytest


def test_oracle_jacobians_match_finite_differences(wca):
    from oracle import kin_spec as ks
    m = wca.synth.icub_like_model()
    b = wca.synth.synth_kin_batch(6, seed=5)
    for i in range(6):
        J = ks.jacobians(m, b["base"][i], b["q"][i])
        N = ks.numeric_jacobians(m, b["base"][i], b["q"][i])
        for k in N:
            assert np.abs(J[k] - N[k]).max() < 5e-9, k
        # mixed representation: the base block is [I -S(p); 0 I]
        assert np.array_equal(J["J_left"][:3, :3], np.eye(3)) and np.array_equal(J["J_left"][3:, 3:6], np.eye(3))
        assert np.array_equal(J["J_left"][3:, :3], np.zeros((3, 3)))
        # a leg joint does not move the other foot, an arm joint moves neither foot nor neck
        assert not J["J_left"][:, 6 + 17:6 + 23].any() and not J["J_right"][:, 6 + 11:6 + 17].any()
        assert not J["J_left"][:, 6 + 3:6 + 11].any() and not J["J_neck"][:, 6 + 3:].any()
        assert J["J_com"][:, 6 + 3:6 + 11].any()              # but it moves the CoM


def test_kin_batches_are_shard_invariant(wca):
    full = wca.synth.synth_kin_batch(9, seed=2)
    part = wca.synth.synth_kin_batch(4, seed=2, first=5)
    assert np.array_equal(full["q"][5:], part["q"]) and np.array_equal(full["base"][5:], part["base"])


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [1, 2, 7, 300])
def test_kinematics_kernel_matches_oracle(wca, batch):
    from oracle import kin_spec as ks
    m = wca.synth.icub_like_model()
    b = wca.synth.synth_kin_batch(batch, seed=9)
    state0 = np.arange(batch * 87, dtype=float).reshape(batch, 87)
    out = wca.KinModel(m).jacobians_host(b["base"], b["q"], state0)
    for i in range(batch):
        r = ks.jacobians(m, b["base"][i], b["q"][i])
        for k in ("J_left", "J_right", "J_neck", "J_com"):
            assert np.abs(out[k][i] - r[k]).max() <= 1e-13, (k, i)
        s = out["state"][i]
        assert np.abs(s[0:3] - r["p_left"]).max() <= 1e-13 and np.abs(s[3:12] - r["R_left"].reshape(9)).max() <= 1e-13
        assert np.abs(s[12:15] - r["p_right"]).max() <= 1e-13 and np.abs(s[15:24] - r["R_right"].reshape(9)).max() <= 1e-13
        assert np.abs(s[48:57] - r["R_neck"].reshape(9)).max() <= 1e-13 and np.abs(s[66:69] - r["com"]).max() <= 1e-13
        untouched = np.r_[24:48, 57:66, 69:87]
        assert np.array_equal(s[untouched], state0[i][untouched])          # desired entries are left alone


@pytest.mark.gpu
def test_kinematics_feeds_the_ik_kernel(wca, qs):
    """Kinematics -> IK: Jacobians and actual poses from the kinematics kernel, desired poses a small step away;
    the IK optimum must be the exact optimum of the QP assembled from the ORACLE's Jacobians."""
    from oracle import kin_spec as ks
    B = 64
    m = wca.synth.icub_like_model()
    kb = wca.synth.synth_kin_batch(B, seed=12)
    ib = wca.synth.synth_ik_batch(B, seed=13)             # twists / CoM velocity references come from here
    out = wca.KinModel(m).jacobians_host(kb["base"], kb["q"], ib["state"])
    s = out["state"]
    # desired = actual shifted a little, so that the correction terms are small and the QP stays feasible
    s[:, 24:36] = s[:, 0:12]; s[:, 24:27] += 0.002; s[:, 36:48] = s[:, 12:24]; s[:, 57:66] = s[:, 48:57]
    s[:, 69:72] = s[:, 66:69] + 0.001
    sol = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=1.5).solve_host(out["J_left"], out["J_right"], out["J_neck"], out["J_com"], kb["q"], s)
    p = qs.IKParams(v_max=1.5 * np.ones(23))
    n_ok = 0
    for i in range(B):
        r = ks.jacobians(m, kb["base"][i], kb["q"][i])
        x = qs.ik_inputs_from_batch(dict(J_left=r["J_left"][None], J_right=r["J_right"][None], J_neck=r["J_neck"][None],
                                         J_com=r["J_com"][None], q=kb["q"][i][None], state=s[i][None]), 0)
        try:
            e = qs.ik_exact(p, x, "qpoases")
        except qs.QPOracleError:
            assert sol["status"][i] != 0
            continue
        assert sol["status"][i] == 0 and np.abs(sol["dq"][i] - e["dq"]).max() <= 1e-8
        n_ok += 1
    assert n_ok > B // 2
