// Minimal value types standing in for the iDynTree / YARP types that appear in the
// reference's solver interfaces (iDynTree and YARP are not available in this image):
//   iDynTree::Vector2/Vector3/Position/Rotation/Transform/Twist/VectorDynSize/MatrixDynSize,
//   yarp::os::Searchable (key -> number | nested list), deg2rad.
// Only what WalkingController / WalkingQPIK touch is provided; semantics follow iDynTree
// (row-major matrices, Rotation::inverse() = transpose, Transform = {Position, Rotation}).
#pragma once
#include <array>
#include <cmath>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace wc {

struct Vector2 { double v[2]{0, 0}; double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } static constexpr int size() { return 2; } };
struct Vector3 { double v[3]{0, 0, 0}; double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };
using Position = Vector3;

struct Rotation {
    double m[9]{1, 0, 0, 0, 1, 0, 0, 0, 1};
    double& operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }
    Rotation inverse() const { Rotation t; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) t(r, c) = (*this)(c, r); return t; }
    Rotation operator*(const Rotation& o) const {
        Rotation t;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += (*this)(r, k) * o(k, c); t(r, c) = s; }
        return t;
    }
    static Rotation RotZ(double a) { Rotation t; t(0, 0) = std::cos(a); t(0, 1) = -std::sin(a); t(1, 0) = std::sin(a); t(1, 1) = std::cos(a); return t; }
    static Rotation Identity() { return Rotation(); }
    double yaw() const { return std::atan2((*this)(1, 0), (*this)(0, 0)); }   // asRPY()(2)
};

struct Transform {
    Position p; Rotation R;
    const Position& getPosition() const { return p; }
    const Rotation& getRotation() const { return R; }
    void setPosition(const Position& q) { p = q; }
    void setRotation(const Rotation& q) { R = q; }
};

struct Twist { double v[6]{0, 0, 0, 0, 0, 0}; double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };

class VectorDynSize {
    std::vector<double> d_;
public:
    VectorDynSize() = default;
    explicit VectorDynSize(size_t n) : d_(n, 0.0) {}
    size_t size() const { return d_.size(); }
    void resize(size_t n) { d_.assign(n, 0.0); }
    double& operator()(size_t i) { return d_[i]; }
    double operator()(size_t i) const { return d_[i]; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }
};

class MatrixDynSize {                 // row-major, like iDynTree::MatrixDynSize
    std::vector<double> d_; size_t r_ = 0, c_ = 0;
public:
    MatrixDynSize() = default;
    MatrixDynSize(size_t r, size_t c) : d_(r * c, 0.0), r_(r), c_(c) {}
    void resize(size_t r, size_t c) { d_.assign(r * c, 0.0); r_ = r; c_ = c; }
    size_t rows() const { return r_; }
    size_t cols() const { return c_; }
    double& operator()(size_t r, size_t c) { return d_[r * c_ + c]; }
    double operator()(size_t r, size_t c) const { return d_[r * c_ + c]; }
    double* data() { return d_.data(); }
    const double* data() const { return d_.data(); }
};

inline double deg2rad(double d) { return d * 3.14159265358979323846 / 180.0; }

// ---- yarp::os::Searchable stand-in ------------------------------------------------------
// A value is a number, a word, or a (possibly nested) list, in YARP's .ini syntax:
//     key value            key (v0 v1 ...)            key ((r, c, v), (r, c, v))
struct Value {
    enum Kind { Null, Number, Word, List } kind = Null;
    double num = 0.0;
    std::string word;
    std::vector<Value> list;
    bool isNull() const { return kind == Null; }
    bool isList() const { return kind == List; }
    bool isDouble() const { return kind == Number; }
    double asDouble() const { return num; }
    int asInt() const { return (int)num; }
    bool asBool() const { return kind == Number ? num != 0.0 : (word == "true" || word == "1"); }
    size_t size() const { return list.size(); }
    const Value& get(size_t i) const { static const Value null; return i < list.size() ? list[i] : null; }
};

class Searchable {
    std::map<std::string, Value> kv_;
public:
    // parses the text of ONE group of a reference-format .ini ("key value" lines, '#' comments,
    // lists may span lines until their parentheses balance, commas are separators)
    bool fromConfigText(const std::string& text);
    void put(const std::string& key, const Value& v) { kv_[key] = v; }
    void put(const std::string& key, double v) { Value x; x.kind = Value::Number; x.num = v; kv_[key] = x; }
    bool isNull() const { return kv_.empty(); }
    bool check(const std::string& key) const { return kv_.count(key) != 0; }
    const Value& find(const std::string& key) const { static const Value null; auto it = kv_.find(key); return it == kv_.end() ? null : it->second; }
    Value check(const std::string& key, const Value& fallback) const { auto it = kv_.find(key); return it == kv_.end() ? fallback : it->second; }
};

inline Value number(double v) { Value x; x.kind = Value::Number; x.num = v; return x; }

}  // namespace wc
