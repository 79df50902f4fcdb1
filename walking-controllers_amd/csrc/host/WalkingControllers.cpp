// Host mirror of WalkingController / WalkingQPIK over the C ABI.  See WalkingControllers.hpp.
// Citations are relative to /root/reference/modules/Walking_module ("WM/").
#include "WalkingControllers.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace wc {

// ------------------------------------------------------------------ config text parser --
namespace {

struct Tok { enum K { Open, Close, Atom } k; std::string s; };

std::vector<Tok> tokenize(const std::string& t) {
    std::vector<Tok> out;
    size_t i = 0;
    while (i < t.size()) {
        const char c = t[i];
        if (std::isspace((unsigned char)c) || c == ',') { ++i; continue; }
        if (c == '(') { out.push_back({Tok::Open, "("}); ++i; continue; }
        if (c == ')') { out.push_back({Tok::Close, ")"}); ++i; continue; }
        if (c == '"') { size_t j = t.find('"', i + 1); if (j == std::string::npos) j = t.size(); out.push_back({Tok::Atom, t.substr(i + 1, j - i - 1)}); i = j + 1; continue; }
        size_t j = i;
        while (j < t.size() && !std::isspace((unsigned char)t[j]) && t[j] != ',' && t[j] != '(' && t[j] != ')') ++j;
        out.push_back({Tok::Atom, t.substr(i, j - i)});
        i = j;
    }
    return out;
}

Value atom(const std::string& s) {
    Value v;
    char* end = nullptr;
    const double d = std::strtod(s.c_str(), &end);
    if (end && *end == '\0' && !s.empty()) { v.kind = Value::Number; v.num = d; }
    else { v.kind = Value::Word; v.word = s; }
    return v;
}

Value parseSeq(const std::vector<Tok>& tk, size_t& i) {      // after '(' ... until matching ')'
    Value v; v.kind = Value::List;
    while (i < tk.size() && tk[i].k != Tok::Close) {
        if (tk[i].k == Tok::Open) { ++i; v.list.push_back(parseSeq(tk, i)); }
        else { v.list.push_back(atom(tk[i].s)); ++i; }
    }
    if (i < tk.size()) ++i;   // consume ')'
    return v;
}

}  // namespace

bool Searchable::fromConfigText(const std::string& text) {
    std::string pending;
    int depth = 0;
    size_t pos = 0;
    auto flush = [&](const std::string& stmt) {
        const std::vector<Tok> tk = tokenize(stmt);
        if (tk.empty() || tk[0].k != Tok::Atom) return;
        const std::string key = tk[0].s;
        if (key[0] == '[') return;                              // [GROUP] / [include ...] headers
        size_t i = 1;
        Value v;
        if (tk.size() == 1) { v.kind = Value::Number; v.num = 1.0; }           // bare flag, e.g. "use_mpc"
        else if (tk.size() == 2 && tk[1].k == Tok::Atom) v = atom(tk[1].s);
        else if (tk[1].k == Tok::Open) { ++i; v = parseSeq(tk, i); }
        else { v.kind = Value::List; for (; i < tk.size(); ++i) if (tk[i].k == Tok::Atom) v.list.push_back(atom(tk[i].s)); }
        kv_[key] = v;
    };
    while (pos <= text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        std::string line = text.substr(pos, eol - pos);
        pos = eol + 1;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        for (char c : line) { if (c == '(') ++depth; else if (c == ')') --depth; }
        pending += line + " ";
        if (depth <= 0) { flush(pending); pending.clear(); depth = 0; }
    }
    return !kv_.empty();
}

namespace {

bool getNumber(const Searchable& cfg, const std::string& key, double& out) {     // YarpHelper::getNumberFromSearchable
    const Value& v = cfg.find(key);
    if (v.isNull() || !v.isDouble()) { std::fprintf(stderr, "[getNumberFromSearchable] Missing field %s\n", key.c_str()); return false; }
    out = v.asDouble();
    return true;
}

bool listToVector(const Value& v, std::vector<double>& out, size_t expect) {     // yarpListToiDynTreeVectorDynSize
    if (v.isNull() || !v.isList() || v.size() != expect) return false;
    out.resize(expect);
    for (size_t i = 0; i < expect; ++i) { if (!v.get(i).isDouble()) return false; out[i] = v.get(i).asDouble(); }
    return true;
}

// iDynTreeHelper::Triplets::getTripletsFromValues builds Triplet(col, row, v) — it transposes
// (UT/src/Utils.cpp:101); reproduced here.
bool tripletsToDense(const Value& v, int dim, double* M) {
    if (v.isNull() || !v.isList()) return false;
    std::fill(M, M + dim * dim, 0.0);
    for (size_t i = 0; i < v.size(); ++i) {
        const Value& t = v.get(i);
        if (!t.isList() || t.size() != 3) { std::fprintf(stderr, "[getSparseMatrixFromTriplets] The triplet must have three elements.\n"); return false; }
        const int row = t.get(0).asInt(), col = t.get(1).asInt();
        if (row >= dim || col >= dim || row < 0 || col < 0) { std::fprintf(stderr, "[getSparseMatrixFromTriplets] element position exceeds the matrix dimension.\n"); return false; }
        M[col * dim + row] = t.get(2).asDouble();
    }
    return true;
}

}  // namespace

// -------------------------------------------------------------------- convex hull helper --
bool ConvexHullHelper::buildConvexHull(const std::vector<std::array<double, 8>>& rects, const std::vector<Transform>& tf) {
    if (rects.size() != tf.size() || rects.empty()) return false;
    std::vector<std::array<double, 2>> pts;
    for (size_t f = 0; f < rects.size(); ++f)
        for (int k = 0; k < 4; ++k) {
            const double x = rects[f][2 * k], y = rects[f][2 * k + 1];
            const Rotation& R = tf[f].getRotation();
            const Position& p = tf[f].getPosition();
            // foot-frame corner (x, y, 0) -> world, projected on the XY plane through the origin
            pts.push_back({R(0, 0) * x + R(0, 1) * y + p(0), R(1, 0) * x + R(1, 1) * y + p(1)});
        }
    std::sort(pts.begin(), pts.end());
    auto cross = [](const std::array<double, 2>& o, const std::array<double, 2>& a, const std::array<double, 2>& b) {
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0]);
    };
    std::vector<std::array<double, 2>> lo, up;
    for (const auto& p : pts) { while (lo.size() >= 2 && cross(lo[lo.size() - 2], lo.back(), p) <= 0) lo.pop_back(); lo.push_back(p); }
    for (auto it = pts.rbegin(); it != pts.rend(); ++it) { while (up.size() >= 2 && cross(up[up.size() - 2], up.back(), *it) <= 0) up.pop_back(); up.push_back(*it); }
    lo.pop_back(); up.pop_back();
    lo.insert(lo.end(), up.begin(), up.end());                 // CCW vertices
    const size_t nc = lo.size();
    if (nc < 3) return false;
    A.resize(nc, 2); b.resize(nc);
    for (size_t k = 0; k < nc; ++k) {
        const auto& v0 = lo[k]; const auto& v1 = lo[(k + 1) % nc];
        const double dx = v1[0] - v0[0], dy = v1[1] - v0[1], len = std::hypot(dx, dy);
        A(k, 0) = dy / len; A(k, 1) = -dx / len;
        b(k) = A(k, 0) * v0[0] + A(k, 1) * v0[1];
    }
    return true;
}

double ConvexHullHelper::computeMargin(const Vector2& u) const {
    double m = 1e300;
    for (size_t k = 0; k < A.rows(); ++k) m = std::min(m, b(k) - A(k, 0) * u(0) - A(k, 1) * u(1));   // unit normals
    return m;
}

// ------------------------------------------------------------------- WalkingController ----
WalkingController::~WalkingController() { if (m_handle) wcqp_mpc_destroy(m_handle); }

bool WalkingController::initialize(const Searchable& config) {
    // …PredictiveController.cpp:311-362 (initialize), :170-243 (initializeMatrices), :245-309 (initializeConstraints)
    const Value& input = config.find("initial_zmp_position");
    if (input.isNull()) { std::fprintf(stderr, "[initialize] Empty initial zmp position.\n"); return false; }
    if (!input.isList() || input.size() != 2) { std::fprintf(stderr, "[initialize] The dimension set in the configuration file is not 2.\n"); return false; }
    for (int i = 0; i < 2; ++i) {
        if (!input.get(i).isDouble()) { std::fprintf(stderr, "[initialize] The zmp position is expected to be a double\n"); return false; }
        m_output(i) = input.get(i).asDouble();
    }
    if (config.isNull()) { std::fprintf(stderr, "[initialize] Empty configuration for walking controller.\n"); return false; }
    const double dT = config.check("sampling_time", number(0.016)).asDouble();
    const double horizonSeconds = config.check("controllerHorizon", number(2.0)).asDouble();
    m_controllerHorizon = (int)std::lround(horizonSeconds / dT);
    wcqp_mpc_params p{};
    p.horizon = m_controllerHorizon;
    p.sampling_time = dT;
    if (!tripletsToDense(config.find("stateWeightTriplets"), 2, p.Q)) { std::fprintf(stderr, "Initialization failed while reading stateWeightTriplets vector.\n"); return false; }
    if (!tripletsToDense(config.find("inputWeightTriplets"), 2, p.R)) { std::fprintf(stderr, "Initialization failed while reading inputWeightTriplets vector.\n"); return false; }
    if (!getNumber(config, "com_height", p.com_height)) { std::fprintf(stderr, "[initialize] Unable to get the double from searchable.\n"); return false; }
    p.gravity = config.check("gravity_acceleration", number(9.81)).asDouble();
    // initializeConstraints
    const Value& feet = config.find("foot_size");
    if (feet.isNull() || !feet.isList()) { std::fprintf(stderr, "Please set the foot_size in the configuration file.\n"); return false; }
    if (feet.size() != 2) { std::fprintf(stderr, "Error while reading the feet dimensions. Wrong number of elements.\n"); return false; }
    const Value& xl = feet.get(0); const Value& yl = feet.get(1);
    if (!xl.isList() || xl.size() != 2) { std::fprintf(stderr, "Error while reading the X limits.\n"); return false; }
    if (!yl.isList() || yl.size() != 2) { std::fprintf(stderr, "Error while reading the Y limits.\n"); return false; }
    const double x1 = xl.get(0).asDouble(), x2 = xl.get(1).asDouble(), y1 = yl.get(0).asDouble(), y2 = yl.get(1).asDouble();
    // Polygon::XYRectangleFromOffsets(front, back, left, right), cpp:297-300
    const double front = std::fabs(std::max(x1, x2)), back = std::fabs(std::min(x1, x2));
    const double left = std::fabs(std::max(y1, y2)), right = std::fabs(std::min(y1, y2));
    m_footRectangle = {front, left, front, -right, -back, -right, -back, left};
    m_convexHullTolerance = config.check("convex_hull_tolerance", number(0.01)).asDouble();
    p.convex_hull_tolerance = m_convexHullTolerance;
    p.feas_tol = 0.0;
    m_params = p;
    if (m_handle) { wcqp_mpc_destroy(m_handle); m_handle = nullptr; }
    const int rc = wcqp_mpc_create(&p, &m_handle);
    if (rc != WCQP_OK) { std::fprintf(stderr, "[initialize] Error while the matrices are initialized: %s\n", wcqp_strerror(rc)); return false; }
    m_refWindow.assign((size_t)(m_controllerHorizon + 1) * 2, 0.0);
    reset();
    return true;
}

bool WalkingController::setConvexHullConstraint(const std::deque<Transform>& leftFoot, const std::deque<Transform>& rightFoot,
                                                const std::deque<bool>& leftInContact, const std::deque<bool>& rightInContact) {
    // …PredictiveController.cpp:364-435
    if (leftFoot.empty() || rightFoot.empty() || leftInContact.empty() || rightInContact.empty()) return false;
    const auto feetStatus = std::make_pair(leftInContact.front(), rightInContact.front());
    if (m_feetStatus == feetStatus) return true;             // hull already evaluated: do nothing (:369-374)
    m_feetStatus = feetStatus;
    bool ok = true;
    if (feetStatus == std::make_pair(true, true)) ok = m_convexHullComputer.buildConvexHull({m_footRectangle, m_footRectangle}, {leftFoot.front(), rightFoot.front()});
    else if (feetStatus == std::make_pair(true, false)) ok = m_convexHullComputer.buildConvexHull({m_footRectangle}, {leftFoot.front()});
    else if (feetStatus == std::make_pair(false, true)) ok = m_convexHullComputer.buildConvexHull({m_footRectangle}, {rightFoot.front()});
    else { std::fprintf(stderr, "[setConvexHullConstraint] None foot is in contact How is it possible?.\n"); return false; }
    if (!ok) { std::fprintf(stderr, "[setConvexHullConstraint] Error while the contraints are evaluated.\n"); return false; }
    if (m_convexHullComputer.A.rows() > WCQP_HULL_ROWS) { std::fprintf(stderr, "[setConvexHullConstraint] more than %d hull rows.\n", WCQP_HULL_ROWS); return false; }
    // the reference allocates a brand-new MPCSolver here (:415-420): cold start, gradient rebuilt
    m_solverInitialized = false;
    return true;
}

bool WalkingController::setFeedback(const Vector2& currentState) {
    if (m_convexHullComputer.A.rows() == 0) { std::fprintf(stderr, "[setFeedback] setConvexHullConstraint has not been called.\n"); return false; }
    m_feedback = currentState;                                // MPCSolver::setBounds rows 0..1 (MPCSolver.cpp:143-146)
    return true;
}

bool WalkingController::setReferenceSignal(const std::deque<Vector2>& referenceSignal, const bool& resetTrajectory) {
    // MPCSolver::setGradient (MPCSolver.cpp:183-239) expressed on the reference window itself:
    // q_x[i] = -Q r_i is a bijection of the window because Q is fixed.
    if (referenceSignal.empty()) return false;
    const int N = m_controllerHorizon;
    if (!m_solverInitialized || resetTrajectory) {
        for (int i = 0; i <= N; ++i) {
            const Vector2& r = (size_t)i < referenceSignal.size() ? referenceSignal[i] : referenceSignal.back();   // :200-214
            m_refWindow[2 * i] = r(0); m_refWindow[2 * i + 1] = r(1);
        }
    } else {
        std::memmove(m_refWindow.data(), m_refWindow.data() + 2, sizeof(double) * 2 * N);                         // :219-222
        const Vector2& r = referenceSignal.size() >= (size_t)N + 1 ? referenceSignal[N] : referenceSignal.back(); // :224-238
        m_refWindow[2 * N] = r(0); m_refWindow[2 * N + 1] = r(1);
    }
    return true;
}

bool WalkingController::solve() {
    // …PredictiveController.cpp:491-521
    m_isSolutionEvaluated = false;
    if (!m_handle || m_convexHullComputer.A.rows() == 0) { std::fprintf(stderr, "[solve] Unable to initialize the solver.\n"); return false; }
    m_solverInitialized = true;                               // lazy initSolver() (:494-501)
    double hA[WCQP_HULL_ROWS * 2] = {0}, hb[WCQP_HULL_ROWS];
    for (int k = 0; k < WCQP_HULL_ROWS; ++k) hb[k] = 1e30;
    const int32_t nc = (int32_t)m_convexHullComputer.A.rows();
    for (int k = 0; k < nc; ++k) { hA[2 * k] = m_convexHullComputer.A(k, 0); hA[2 * k + 1] = m_convexHullComputer.A(k, 1); hb[k] = m_convexHullComputer.b(k); }
    double u0[2] = {0, 0}, margin = 0;
    int32_t status = 0;
    uint32_t active = 0;
    const int rc = wcqp_mpc_solve_host(m_handle, 1, m_feedback.v, m_refWindow.data(), m_controllerHorizon + 1, m_output.v,
                                       hA, hb, &nc, u0, &status, &active, &margin);
    if (rc != WCQP_OK) { std::fprintf(stderr, "[solve] Unable to solve the problem: %s\n", wcqp_strerror(rc)); return false; }
    lastStatus = status; lastActive = active;
    if (status == WCQP_STATUS_INFEASIBLE) { std::fprintf(stderr, "[solve] Unable to solve the problem.\n"); return false; }
    m_output(0) = u0[0]; m_output(1) = u0[1];                 // overwritten BEFORE the margin check (:510-511, Appendix B-8)
    if (status == WCQP_STATUS_OUTSIDE_HULL) { std::fprintf(stderr, "[solve] The evaluated ZMP is outside the convexHull.\n"); return false; }
    m_isSolutionEvaluated = true;
    return true;
}

bool WalkingController::getControllerOutput(Vector2& controllerOutput) {
    if (!m_isSolutionEvaluated) { std::fprintf(stderr, "[getControllerOutput] The solution is not evaluated. Please call 'solve()' method.\n"); return false; }
    m_isSolutionEvaluated = false;                            // one-shot read (:525-533)
    controllerOutput = m_output;
    return true;
}

void WalkingController::reset() {
    m_feetStatus = std::make_pair(false, false);              // forces a rebuild at the next tick (:537-543)
    m_isSolutionEvaluated = false;
}

// --------------------------------------------------------------------------- WalkingQPIK --
WalkingQPIK::~WalkingQPIK() {}

bool WalkingQPIK::initializeMatrices(const Searchable& config) {
    // WM/src/WalkingQPInverseKinematics.cpp:25-116
    if (!m_useCoMAsConstraint) {
        if (!tripletsToDense(config.find("comWeightTriplets"), 3, m_comWeight)) { std::fprintf(stderr, "Initialization failed while reading comWeightTriplets vector.\n"); return false; }
    }
    if (!tripletsToDense(config.find("neckWeightTriplets"), 3, m_neckWeight)) { std::fprintf(stderr, "Initialization failed while reading neckWeightTriplets vector.\n"); return false; }
    if (!listToVector(config.find("jointRegularizationWeights"), m_jointRegularizationWeights, m_actuatedDOFs)) { std::fprintf(stderr, "Initialization failed while reading jointRegularizationWeights vector.\n"); return false; }
    m_comJacobian.resize(3, m_numberOfVariables); m_neckJacobian.resize(3, m_numberOfVariables);
    m_leftFootJacobian.resize(6, m_numberOfVariables); m_rightFootJacobian.resize(6, m_numberOfVariables);
    if (!listToVector(config.find("jointRegularizationGains"), m_jointRegularizationGains, m_actuatedDOFs)) { std::fprintf(stderr, "Initialization failed while reading jointRegularizationGains vector.\n"); return false; }
    if (!getNumber(config, "k_posFoot", m_kPosFoot)) { std::fprintf(stderr, "Initialization failed while reading k_posFoot.\n"); return false; }
    if (!getNumber(config, "k_attFoot", m_kAttFoot)) { std::fprintf(stderr, "Initialization failed while reading k_attFoot.\n"); return false; }
    if (!getNumber(config, "k_neck", m_kNeck)) { std::fprintf(stderr, "Initialization failed while reading k_neck.\n"); return false; }
    if (!getNumber(config, "k_posCom", m_kCom)) { std::fprintf(stderr, "Initialization failed while reading k_posCom.\n"); return false; }
    return true;
}

bool WalkingQPIK::setRobotState(const VectorDynSize& jointPosition, const Transform& leftFootToWorldTransform,
                                const Transform& rightFootToWorldTransform, const Rotation& neckOrientation,
                                const Position& comPosition) {
    if ((int)jointPosition.size() != m_actuatedDOFs) { std::fprintf(stderr, "[setRobotState] The size of the jointPosition vector is not coherent with the number of the actuated Joint\n"); return false; }
    m_jointPosition = jointPosition;
    m_leftFootToWorldTransform = leftFootToWorldTransform; m_rightFootToWorldTransform = rightFootToWorldTransform;
    m_neckOrientation = neckOrientation; m_comPosition = comPosition;
    return true;
}

void WalkingQPIK::setDesiredNeckOrientation(const Rotation& desiredNeckOrientation) {
    m_desiredNeckOrientation = desiredNeckOrientation * m_additionalRotation;       // base.cpp:143-146
}

static bool checkJac(const MatrixDynSize& J, size_t rows, int dof, const char* who) {
    if (J.rows() != rows) { std::fprintf(stderr, "[%s] the number of rows has to be equal to %zu.\n", who, rows); return false; }
    if ((int)J.cols() != dof + 6) { std::fprintf(stderr, "[%s] the number of rows has to be equal to %d\n", who, dof + 6); return false; }
    return true;
}
bool WalkingQPIK::setCoMJacobian(const MatrixDynSize& J) { if (!checkJac(J, 3, m_actuatedDOFs, "setCoMJacobian")) return false; m_comJacobian = J; return true; }
bool WalkingQPIK::setLeftFootJacobian(const MatrixDynSize& J) { if (!checkJac(J, 6, m_actuatedDOFs, "setLeftFootJacobian")) return false; m_leftFootJacobian = J; return true; }
bool WalkingQPIK::setRightFootJacobian(const MatrixDynSize& J) { if (!checkJac(J, 6, m_actuatedDOFs, "setRightFootJacobian")) return false; m_rightFootJacobian = J; return true; }
bool WalkingQPIK::setNeckJacobian(const MatrixDynSize& J) {
    if (!checkJac(J, 6, m_actuatedDOFs, "setNeckJacobian")) return false;
    for (int r = 0; r < 3; ++r)                                                      // keeps rows 3..5 (base.cpp:214-215)
        for (int c = 0; c < m_numberOfVariables; ++c) m_neckJacobian(r, c) = J(3 + r, c);
    return true;
}
bool WalkingQPIK::setDesiredJointPosition(const VectorDynSize& regularizationTerm) {
    if ((int)regularizationTerm.size() != m_actuatedDOFs) { std::fprintf(stderr, "[setDesiredJointPosition] The number of the desired joint position has to be equal to the number of actuated joints\n"); return false; }
    m_regularizationTerm = regularizationTerm;
    // the posture enters the gradient of every solve (osqp.cpp:185, qp.cpp:166): hand it to the solver handle
    if (!pushPosture()) return false;
    return true;
}
void WalkingQPIK::setDesiredFeetTransformation(const Transform& l, const Transform& r) { m_desiredLeftFootToWorldTransform = l; m_desiredRightFootToWorldTransform = r; }
void WalkingQPIK::setDesiredFeetTwist(const Twist& l, const Twist& r) { m_leftFootTwist = l; m_rightFootTwist = r; }
void WalkingQPIK::setDesiredCoMVelocity(const Vector3& v) { m_comVelocity = v; }
void WalkingQPIK::setDesiredCoMPosition(const Position& p) { m_desiredComPosition = p; }

// ----------------------------------------------------------------------- WalkingQPIK_hip --
WalkingQPIK_hip::~WalkingQPIK_hip() { if (m_handle) wcqp_ik_destroy(m_handle); }

bool WalkingQPIK_hip::initialize(const Searchable& config, const int& actuatedDOFs,
                                 const VectorDynSize& minJointsLimit, const VectorDynSize& maxJointsLimit) {
    // osqp.cpp:54-133 / qp.cpp:53-133
    m_actuatedDOFs = actuatedDOFs;
    if (config.isNull()) { std::fprintf(stderr, "[initialize] Empty configuration for QP-IK solver.\n"); return false; }
    m_useCoMAsConstraint = config.check("useCoMAsConstraint", number(0)).asBool();
    m_numberOfVariables = m_actuatedDOFs + 6;
    const int task = m_useCoMAsConstraint ? 15 : 12;
    m_numberOfConstraints = m_form == WCQP_IK_FORM_OSQP ? m_actuatedDOFs + task : task;
    if (actuatedDOFs < 1 || actuatedDOFs > WCQP_MAX_DOF) return false;
    std::vector<double> reg;
    if (!listToVector(config.find("jointRegularization"), reg, m_actuatedDOFs)) { std::fprintf(stderr, "[initialize] Unable to convert a YARP list to an iDynTree::VectorDynSize, joint regularization\n"); return false; }
    m_regularizationTerm.resize(m_actuatedDOFs);
    for (int i = 0; i < m_actuatedDOFs; ++i) m_regularizationTerm(i) = deg2rad(reg[i]);          // osqp.cpp:101-102
    m_jointPosition.resize(m_actuatedDOFs);
    if (!initializeMatrices(config)) { std::fprintf(stderr, "[initialize] Unable to Initialize the constant matrix.\n"); return false; }
    if (minJointsLimit.size() != maxJointsLimit.size()) { std::fprintf(stderr, "[setVelocityBounds] The size of the vector limits has to be equal.\n"); return false; }
    if ((int)minJointsLimit.size() != m_actuatedDOFs) { std::fprintf(stderr, "[setVelocityBounds] The size of the vector limits has to be equal to the number of the joint\n"); return false; }
    const Value& rot = config.find("additional_rotation");                                       // iDynTree::parseRotationMatrix
    if (!rot.isList() || rot.size() != 3) { std::fprintf(stderr, "[initialize] Unable to set the additional rotation.\n"); return false; }
    for (int r = 0; r < 3; ++r) {
        if (!rot.get(r).isList() || rot.get(r).size() != 3) { std::fprintf(stderr, "[initialize] Unable to set the additional rotation.\n"); return false; }
        for (int c = 0; c < 3; ++c) m_additionalRotation(r, c) = rot.get(r).get(c).asDouble();
    }
    wcqp_ik_params p{};
    p.dof = m_actuatedDOFs; p.use_com_as_constraint = m_useCoMAsConstraint ? 1 : 0; p.form = m_form; p.max_iter = 100;
    std::memcpy(p.com_weight, m_comWeight, sizeof(m_comWeight));
    std::memcpy(p.neck_weight, m_neckWeight, sizeof(m_neckWeight));
    for (int i = 0; i < m_actuatedDOFs; ++i) {
        p.joint_reg_weights[i] = m_jointRegularizationWeights[i];
        p.joint_reg_gains[i] = m_jointRegularizationGains[i];
        p.joint_reg_rad[i] = m_regularizationTerm(i);
        p.v_min[i] = minJointsLimit(i); p.v_max[i] = maxJointsLimit(i);
    }
    p.k_pos_com = m_kCom; p.k_pos_foot = m_kPosFoot; p.k_att_foot = m_kAttFoot; p.k_neck = m_kNeck;
    // the caller of these classes hands over iDynTree free-floating Jacobians in MIXED representation
    // (WalkingForwardKinematics.cpp:33, 436-454); anything else still works (general kernel behind the check)
    p.jacobian_structure = WCQP_IK_JAC_AUTO;
    if (m_handle) { wcqp_ik_destroy(m_handle); m_handle = nullptr; }
    const int rc = wcqp_ik_create(&p, &m_handle);
    if (rc != WCQP_OK) { std::fprintf(stderr, "[initialize] %s\n", wcqp_strerror(rc)); return false; }
    m_solution.assign(m_actuatedDOFs, 0.0); m_footErr.assign(12, 0.0);
    return true;
}

bool WalkingQPIK_hip::pushPosture() {
    if (!m_handle) return true;                       // before initialize(): picked up when the handle is created
    std::vector<double> q(m_actuatedDOFs);
    for (int i = 0; i < m_actuatedDOFs; ++i) q[i] = m_regularizationTerm(i);
    const int rc = wcqp_ik_set_posture(m_handle, q.data());
    if (rc != WCQP_OK) { std::fprintf(stderr, "[setDesiredJointPosition] %s\n", wcqp_strerror(rc)); return false; }
    return true;
}

bool WalkingQPIK_hip::solve() {
    // osqp.cpp:340-393 / qp.cpp:284-339: assemble + solve happen inside the kernel
    m_isSolutionEvaluated = false;
    if (!m_handle) return false;
    double st[WCQP_IK_STATE_LEN];
    auto putT = [&](int po, int ro, const Transform& T) { for (int k = 0; k < 3; ++k) st[po + k] = T.getPosition()(k); std::memcpy(st + ro, T.getRotation().m, sizeof(double) * 9); };
    putT(0, 3, m_leftFootToWorldTransform); putT(12, 15, m_rightFootToWorldTransform);
    putT(24, 27, m_desiredLeftFootToWorldTransform); putT(36, 39, m_desiredRightFootToWorldTransform);
    std::memcpy(st + 48, m_neckOrientation.m, sizeof(double) * 9);
    std::memcpy(st + 57, m_desiredNeckOrientation.m, sizeof(double) * 9);
    for (int k = 0; k < 3; ++k) { st[66 + k] = m_comPosition(k); st[69 + k] = m_desiredComPosition(k); st[72 + k] = m_comVelocity(k); }
    for (int k = 0; k < 6; ++k) { st[75 + k] = m_leftFootTwist(k); st[81 + k] = m_rightFootTwist(k); }
    int32_t status = 0, iters = 0;
    uint32_t lo = 0, up = 0;
    const int rc = wcqp_ik_solve_host(m_handle, 1, m_leftFootJacobian.data(), m_rightFootJacobian.data(),
                                      m_neckJacobian.data(), m_comJacobian.data(), m_jointPosition.data(), st,
                                      m_solution.data(), &status, &lo, &up, m_footErr.data(), &iters);
    if (rc != WCQP_OK) { std::fprintf(stderr, "[solve] Unable to solve the problem: %s\n", wcqp_strerror(rc)); return false; }
    m_status = status; m_activeLower = lo; m_activeUpper = up;
    if (status != WCQP_STATUS_SOLVED) { std::fprintf(stderr, "[solve] Unable to solve the problem.\n"); return false; }
    m_isSolutionEvaluated = true;
    return true;
}

bool WalkingQPIK_hip::getSolution(VectorDynSize& output) {
    if (!m_isSolutionEvaluated) { std::fprintf(stderr, "[getSolution] The solution is not evaluated. Please call 'solve()' method.\n"); return false; }
    if ((int)output.size() != m_actuatedDOFs) output.resize(m_actuatedDOFs);
    for (int i = 0; i < m_actuatedDOFs; ++i) output(i) = m_solution[i];
    if (m_form == WCQP_IK_FORM_QPOASES) m_isSolutionEvaluated = false;       // qp.cpp:360 clears the flag, osqp.cpp:410-428 does not (B-17)
    return true;
}

bool WalkingQPIK_hip::getLeftFootError(VectorDynSize& output) {
    if (m_form == WCQP_IK_FORM_OSQP && !m_isSolutionEvaluated) { std::fprintf(stderr, "[getLeftFootError] The solution is not evaluated. Please call 'solve()' method.\n"); return false; }
    if (output.size() != 6) output.resize(6);
    for (int k = 0; k < 6; ++k) output(k) = m_footErr[k];
    return true;
}

bool WalkingQPIK_hip::getRightFootError(VectorDynSize& output) {
    if (m_form == WCQP_IK_FORM_OSQP && !m_isSolutionEvaluated) { std::fprintf(stderr, "[getRightFootError] The solution is not evaluated. Please call 'solve()' method.\n"); return false; }
    if (output.size() != 6) output.resize(6);
    for (int k = 0; k < 6; ++k) output(k) = m_footErr[6 + k];
    return true;
}

}  // namespace wc
