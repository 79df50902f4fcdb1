// C++ host mirror of the reference's two solver interfaces, on top of the C ABI (wcqp.h).
//
//   class WalkingController  — same public methods, argument order and bool semantics as
//       WM/include/WalkingDCMModelPredictiveController.hpp:190-250 (the class is called
//       WalkingController although the file is WalkingDCMModelPredictiveController.*).
//   class WalkingQPIK        — abstract base with the reference's setters
//       (WM/include/WalkingQPInverseKinematics.hpp:81-208);
//   class WalkingQPIK_hip    — the concrete back-end that replaces WalkingQPIK_osqp and
//       WalkingQPIK_qpOASES; `form` selects which of the two reference back-ends'
//       formulation is reproduced.  WalkingQPIK_osqp / WalkingQPIK_qpOASES are provided as
//       thin subclasses so that WalkingModule::configure (WM/src/WalkingModule.cpp:233-258)
//       compiles unchanged.
//
// One object = one robot (batch of 1 through the C ABI), exactly like the reference; the
// batched path is the C ABI itself.  iDynTree / YARP types are replaced by wc_types.hpp.
// Errors: every method returns bool, diagnostics go to stderr, nothing throws.
#pragma once
#include <deque>
#include <utility>
#include "wc_types.hpp"
#include "wcqp.h"

namespace wc {

// Support-polygon helper (stands in for iDynTree::ConvexHullProjectionConstraint, whose
// row order/normalisation is upstream and unpinned — SURVEY Appendix D-4).  Convention:
// CCW hull of the feet rectangles projected on the XY plane, unit outward normals,
// rows A u <= b; computeMargin = signed distance to the boundary, positive inside.
struct ConvexHullHelper {
    MatrixDynSize A;       // nc x 2
    VectorDynSize b;       // nc
    bool buildConvexHull(const std::vector<std::array<double, 8>>& footRectanglesXY /* 4 corners each */,
                         const std::vector<Transform>& transforms);
    double computeMargin(const Vector2& u) const;
};

class WalkingController {
    wcqp_mpc_t m_handle{nullptr};
    wcqp_mpc_params m_params{};
    int m_controllerHorizon{0};
    double m_convexHullTolerance{0.01};
    std::pair<bool, bool> m_feetStatus{false, false};
    ConvexHullHelper m_convexHullComputer;
    std::array<double, 8> m_footRectangle{};         // corners (x,y) x 4 in the foot frame
    bool m_isSolutionEvaluated{false};
    bool m_solverInitialized{false};                 // mirrors MPCSolver::isInitialized()
    Vector2 m_output;
    Vector2 m_feedback;
    std::vector<double> m_refWindow;                 // (N+1) x 2, what MPCSolver::m_gradient encodes
public:
    ~WalkingController();
    bool initialize(const Searchable& config);
    bool setConvexHullConstraint(const std::deque<Transform>& leftFoot, const std::deque<Transform>& rightFoot,
                                 const std::deque<bool>& leftInContact, const std::deque<bool>& rightInContact);
    bool setFeedback(const Vector2& currentState);
    bool setReferenceSignal(const std::deque<Vector2>& referenceSignal, const bool& resetTrajectory);
    bool solve();
    bool getControllerOutput(Vector2& controllerOutput);
    void reset();
    // introspection for tests
    const ConvexHullHelper& hull() const { return m_convexHullComputer; }
    int lastStatus{0};
    unsigned lastActive{0};
};

class WalkingQPIK {
protected:
    MatrixDynSize m_comJacobian, m_neckJacobian, m_leftFootJacobian, m_rightFootJacobian;
    Twist m_leftFootTwist, m_rightFootTwist;
    Vector3 m_comVelocity;
    Position m_desiredComPosition;
    Transform m_desiredLeftFootToWorldTransform, m_desiredRightFootToWorldTransform;
    Rotation m_desiredNeckOrientation, m_additionalRotation;
    VectorDynSize m_regularizationTerm;
    Position m_comPosition;
    Transform m_leftFootToWorldTransform, m_rightFootToWorldTransform;
    Rotation m_neckOrientation;
    VectorDynSize m_jointPosition;
    int m_numberOfVariables{0}, m_numberOfConstraints{0}, m_actuatedDOFs{0};
    std::vector<double> m_jointRegularizationGains, m_jointRegularizationWeights;
    double m_kPosFoot{0}, m_kAttFoot{0}, m_kNeck{0}, m_kCom{0};
    double m_comWeight[9]{}, m_neckWeight[9]{};
    bool m_isSolutionEvaluated{false};
    bool m_useCoMAsConstraint{false};
    virtual bool initializeMatrices(const Searchable& config);
    virtual bool pushPosture() { return true; }      // back-end hook of setDesiredJointPosition
public:
    virtual ~WalkingQPIK();
    virtual bool initialize(const Searchable& config, const int& actuatedDOFs,
                            const VectorDynSize& minJointsLimit, const VectorDynSize& maxJointsLimit) = 0;
    bool setRobotState(const VectorDynSize& jointPosition, const Transform& leftFootToWorldTransform,
                       const Transform& rightFootToWorldTransform, const Rotation& neckOrientation,
                       const Position& comPosition);
    bool setCoMJacobian(const MatrixDynSize& comJacobian);
    bool setLeftFootJacobian(const MatrixDynSize& leftFootJacobian);
    bool setRightFootJacobian(const MatrixDynSize& rightFootJacobian);
    bool setNeckJacobian(const MatrixDynSize& neckJacobian);
    bool setDesiredJointPosition(const VectorDynSize& regularizationTerm);
    void setDesiredFeetTwist(const Twist& leftFootTwist, const Twist& rightFootTwist);
    void setDesiredCoMVelocity(const Vector3& comVelocity);
    void setDesiredFeetTransformation(const Transform& desiredLeftFootToWorldTransform,
                                      const Transform& desiredRightFootToWorldTransform);
    void setDesiredNeckOrientation(const Rotation& desiredNeckOrientation);
    void setDesiredCoMPosition(const Position& desiredComPosition);
    virtual bool solve() = 0;
    virtual bool getSolution(VectorDynSize& output) = 0;
    virtual bool getLeftFootError(VectorDynSize& output) = 0;
    virtual bool getRightFootError(VectorDynSize& output) = 0;
};

class WalkingQPIK_hip : public WalkingQPIK {
    wcqp_ik_t m_handle{nullptr};
    int m_form;
    std::vector<double> m_solution, m_footErr;
    unsigned m_activeLower{0}, m_activeUpper{0};
    int m_status{0};
public:
    explicit WalkingQPIK_hip(int form = WCQP_IK_FORM_QPOASES) : m_form(form) {}
    ~WalkingQPIK_hip() override;
    bool initialize(const Searchable& config, const int& actuatedDOFs,
                    const VectorDynSize& minJointsLimit, const VectorDynSize& maxJointsLimit) final;
    bool pushPosture() final;
    bool solve() final;
    bool getSolution(VectorDynSize& output) final;
    bool getLeftFootError(VectorDynSize& output) final;
    bool getRightFootError(VectorDynSize& output) final;
    unsigned activeLower() const { return m_activeLower; }
    unsigned activeUpper() const { return m_activeUpper; }
    int status() const { return m_status; }
};

// drop-in names of the reference's two back-ends
struct WalkingQPIK_osqp : WalkingQPIK_hip { WalkingQPIK_osqp() : WalkingQPIK_hip(WCQP_IK_FORM_OSQP) {} };
struct WalkingQPIK_qpOASES : WalkingQPIK_hip { WalkingQPIK_qpOASES() : WalkingQPIK_hip(WCQP_IK_FORM_QPOASES) {} };

}  // namespace wc
