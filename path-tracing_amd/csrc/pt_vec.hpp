// pt_vec.hpp -- small vector / quaternion / pose algebra for host and device.
//
// Operation order matters here: results must be bit-identical to the
// reference's vec<N>, quat and Transformation classes (reference
// src/15/math.hpp:21-123,127-242, transformation.hpp:6-86) under IEEE
// arithmetic without contraction, so every function spells out the same
// sequence of additions and multiplications (e.g. dot() starts from 0 and
// adds the products left to right; a quaternion rotates a vector by two full
// Hamilton products).  Build with -ffp-contract=off.
#pragma once

#include "pt_amd.h"
#include "pt_math.hpp"

namespace ptv {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kEps = 1.1920928955078125e-07f; // FLT_EPSILON
constexpr float kFltMax = 3.402823466e+38f;

struct V2 {
    float x, y;
};

struct V3 {
    float x, y, z;
};

PT_HD V3 v3(float a, float b, float c) { return V3{ a, b, c }; }
PT_HD V3 v3(float a) { return V3{ a, a, a }; }
PT_HD V3 v3(const float* p) { return V3{ p[0], p[1], p[2] }; }
PT_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
PT_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PT_HD V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
PT_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
PT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }

PT_HD float mixf(float x, float y, float a) { return x + a * (y - x); }
PT_HD float fractf(float x) { return x - ::floorf(x); }
PT_HD V3 mix3(V3 a, V3 b, float t) { return v3(mixf(a.x, b.x, t), mixf(a.y, b.y, t), mixf(a.z, b.z, t)); }
PT_HD float fabs_(float x) { return __builtin_fabsf(x); }
PT_HD float max_(float a, float b) { return a < b ? b : a; }                  // std::max(a, b)
PT_HD float min_(float a, float b) { return b < a ? b : a; }                  // std::min(a, b)

PT_HD float dot(V3 a, V3 b)
{
    float r = 0.0f;
    r += a.x * b.x;
    r += a.y * b.y;
    r += a.z * b.z;
    return r;
}
PT_HD float dot(V2 a, V2 b)
{
    float r = 0.0f;
    r += a.x * b.x;
    r += a.y * b.y;
    return r;
}
PT_HD float length(V3 a) { return ::sqrtf(dot(a, a)); }
PT_HD float length(V2 a) { return ::sqrtf(dot(a, a)); }
PT_HD V3 normalize(V3 a) { return a / length(a); }
PT_HD V3 cross(V3 v, V3 w) { return v3(v.y * w.z - v.z * w.y, v.z * w.x - v.x * w.z, v.x * w.y - v.y * w.x); }
PT_HD V3 reflect(V3 i, V3 n) { return i - 2.0f * dot(n, i) * n; }
PT_HD V3 refract(V3 i, V3 n, float eta)
{
    const float d = dot(n, i);
    const float k = 1.0f - eta * eta * (1.0f - d * d);
    return k <= 0.0f ? v3(0.0f) : i * eta - n * (eta * d + ::sqrtf(k));
}

struct Quat {
    float x, y, z, w;
};
PT_HD Quat conjugate(Quat q) { return Quat{ -q.x, -q.y, -q.z, q.w }; }
PT_HD Quat qmul(Quat a, Quat q)
{
    Quat p;
    p.x = a.w * q.x + a.x * q.w + a.y * q.z - a.z * q.y;
    p.y = a.w * q.y + a.y * q.w + a.z * q.x - a.x * q.z;
    p.z = a.w * q.z + a.z * q.w + a.x * q.y - a.y * q.x;
    p.w = a.w * q.w - a.x * q.x - a.y * q.y - a.z * q.z;
    return p;
}
PT_HD V3 rotate(Quat q, V3 v)
{
    Quat t = qmul(qmul(q, Quat{ v.x, v.y, v.z, 0.0f }), conjugate(q));
    return v3(t.x, t.y, t.z);
}

struct Xform {
    V3 translation;
    Quat rotation;
    V3 scaling;
};
PT_HD Xform xformIdentity() { return Xform{ v3(0.0f), Quat{ 0.0f, 0.0f, 0.0f, 1.0f }, v3(1.0f) }; }
PT_HD Xform xformOf(const pt_transform& p)
{
    return Xform{ v3(p.translation), Quat{ p.rotation[0], p.rotation[1], p.rotation[2], p.rotation[3] }, v3(p.scaling) };
}
PT_HD V3 apply(const Xform& T, V3 v) { return T.translation + rotate(T.rotation, v * T.scaling); }

// the portable transcendentals (what the device uses everywhere)
struct PortableMath {
    PT_HD float sin(float x) { return ptm::sinf(x); }
    PT_HD float cos(float x) { return ptm::cosf(x); }
    PT_HD float acos(float x) { return ptm::acosf(x); }
};

template <class M>
PT_HD Quat slerp(Quat q, Quat r, float alpha)
{
    Quat w = r;
    float cosHalf = q.x * r.x + q.y * r.y + q.z * r.z + q.w * r.w;
    if (cosHalf < 0.0f) {
        w.x = -w.x;
        w.y = -w.y;
        w.z = -w.z;
        w.w = -w.w;
        cosHalf = -cosHalf;
    }
    float tq, tw;
    if (cosHalf >= 1.0f) {
        tq = 1.0f;
        tw = 0.0f;
    } else {
        float half = M::acos(cosHalf);
        float sinHalf = ::sqrtf(1.0f - cosHalf * cosHalf);
        if (fabs_(sinHalf) < kEps) {
            tq = 0.5f;
            tw = 0.5f;
        } else {
            tq = M::sin((1.0f - alpha) * half) / sinHalf;
            tw = M::sin(alpha * half) / sinHalf;
        }
    }
    return Quat{ q.x * tq + w.x * tw, q.y * tq + w.y * tw, q.z * tq + w.z * tw, q.w * tq + w.w * tw };
}

// Animation::at(t) for the closed set of animation types (reference
// src/15/animation.hpp:8-11, animation_constant.hpp:14-17,
// src/07/pathtracer-frames.cpp:83-87, src/15/transformation.hpp:80-86)
template <class M>
PT_HD Xform animationAt(const pt_animation& a, float t)
{
    if (a.type == PT_ANIM_CONSTANT)
        return xformOf(a.T0);
    if (a.type == PT_ANIM_TWOKEY) {
        Xform T0 = xformOf(a.T0), T1 = xformOf(a.T1);
        float alpha = t / a.duration;
        return Xform{ mix3(T0.translation, T1.translation, alpha), slerp<M>(T0.rotation, T1.rotation, alpha),
                      mix3(T0.scaling, T1.scaling, alpha) };
    }
    return xformIdentity();
}

struct Ray {
    V3 origin, direction, invDirection;
    float time;
};
PT_HD Ray makeRay(V3 o, V3 d, float t) { return Ray{ o, d, v3(1.0f) / d, t }; }
PT_HD V3 rayAt(const Ray& r, float a) { return r.origin + a * r.direction; }

} // namespace ptv
