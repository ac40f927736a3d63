// pt_oracle_math.hpp -- TEST INFRASTRUCTURE (part of the CPU oracle).
//
// "Portable" transcendentals: sin, cos, exp, pow, asin, acos, atan2 for float
// arguments, evaluated in double precision with nothing but IEEE-754 basic
// operations (+ - * / sqrt, floor, comparisons, bit moves) in a fixed order,
// and rounded to float once at the end.  Because every basic operation is
// correctly rounded on both the host CPU and the GPU, these functions return
// the same bits everywhere when compiled without floating-point contraction,
// which is what lets a GPU render be compared bit for bit with a CPU render.
// The result is the correctly rounded float value except when the exact value
// lies within about 1e-9 ulp of a rounding boundary.
//
// The reference itself calls the C library (std::sin, std::pow, ...: reference
// src/15/sampler.hpp:12-80, material_phong.hpp:29,97, material_glass.hpp:28-30,
// surface_sphere.hpp:20-27, envmap_equirect.hpp:17-18, math.hpp:190-198); glibc's
// float functions are within 1 ulp, so the two agree to the last bit for most
// arguments and differ by one ulp for the rest.  The oracle can run with either.
#pragma once

#include <cstdint>
#include <cstring>
#include <cmath>

namespace pmath {

static inline uint64_t bitsOf(double x)
{
    uint64_t u;
    std::memcpy(&u, &x, 8);
    return u;
}

static inline double fromBits(uint64_t u)
{
    double x;
    std::memcpy(&x, &u, 8);
    return x;
}

static const double kPio2Hi = 1.5707963109016418;     // 0x1.921fb50000000p+0
static const double kPio2Lo = 1.5893254773528196e-08; // pi/2 - kPio2Hi
static const double kTwoOverPi = 0.6366197723675814;
static const double kLn2Hi = 0.6931471675634384;      // 0x1.62e42f8000000p-1
static const double kLn2Lo = 1.2996506893889889e-08;  // ln 2 - kLn2Hi
static const double kLog2e = 1.4426950408889634;
static const double kPi = 3.141592653589793;
static const double kPio2 = 1.5707963267948966;
static const double kPio4 = 0.7853981633974483;

// sin(r), |r| <= pi/4 : Taylor through r^17
static inline double sinKernel(double r)
{
    double z = r * r;
    double p = 2.8114572543455206e-15;
    p = p * z + -7.647163731819816e-13;
    p = p * z + 1.6059043836821613e-10;
    p = p * z + -2.505210838544172e-08;
    p = p * z + 2.7557319223985893e-06;
    p = p * z + -0.0001984126984126984;
    p = p * z + 0.008333333333333333;
    p = p * z + -0.16666666666666666;
    return r + r * (z * p);
}

// cos(r), |r| <= pi/4 : Taylor through r^16
static inline double cosKernel(double r)
{
    double z = r * r;
    double p = 4.779477332387385e-14;
    p = p * z + -1.1470745597729725e-11;
    p = p * z + 2.08767569878681e-09;
    p = p * z + -2.755731922398589e-07;
    p = p * z + 2.48015873015873e-05;
    p = p * z + -0.001388888888888889;
    p = p * z + 0.041666666666666664;
    p = p * z + -0.5;
    return 1.0 + z * p;
}

// quadrant and reduced argument of x
static inline int reducePio2(double x, double& r)
{
    double kd = std::floor(x * kTwoOverPi + 0.5);
    r = (x - kd * kPio2Hi) - kd * kPio2Lo;
    return int(int64_t(kd) & 3);
}

static inline void sincosf(float xf, float& s, float& c)
{
    double x = double(xf);
    if (!(x - x == 0.0)) { // inf or nan
        s = c = float(x - x);
        return;
    }
    if (x == 0.0) { // keeps the sign of a zero argument, as the C library does
        s = xf;
        c = 1.0f;
        return;
    }
    double r;
    int q = reducePio2(x, r);
    double sr = sinKernel(r);
    double cr = cosKernel(r);
    double sd, cd;
    switch (q) {
    case 0: sd = sr; cd = cr; break;
    case 1: sd = cr; cd = -sr; break;
    case 2: sd = -sr; cd = -cr; break;
    default: sd = -cr; cd = sr; break;
    }
    s = float(sd);
    c = float(cd);
}

static inline float sinf(float x)
{
    float s, c;
    sincosf(x, s, c);
    return s;
}

static inline float cosf(float x)
{
    float s, c;
    sincosf(x, s, c);
    return c;
}

// exp(t) for double t; exact enough that float(exp) is correctly rounded;
// results below the float range return 0, above it +inf
static inline double expd(double t)
{
    if (t != t)
        return t;
    if (t > 89.0)
        return HUGE_VAL;
    if (t < -110.0)
        return 0.0;
    double kd = std::floor(t * kLog2e + 0.5);
    double r = (t - kd * kLn2Hi) - kd * kLn2Lo;
    double p = 1.1470745597729725e-11;
    p = p * r + 1.6059043836821613e-10;
    p = p * r + 2.08767569878681e-09;
    p = p * r + 2.505210838544172e-08;
    p = p * r + 2.755731922398589e-07;
    p = p * r + 2.7557319223985893e-06;
    p = p * r + 2.48015873015873e-05;
    p = p * r + 0.0001984126984126984;
    p = p * r + 0.001388888888888889;
    p = p * r + 0.008333333333333333;
    p = p * r + 0.041666666666666664;
    p = p * r + 0.16666666666666666;
    p = p * r + 0.5;
    double e = 1.0 + (r + (r * r) * p);
    int64_t k = int64_t(kd);
    return e * fromBits(uint64_t(k + 1023) << 52);
}

static inline float expf(float x)
{
    return float(expd(double(x)));
}

// ln(x) for positive, finite, normal double x
static inline double logd(double x)
{
    uint64_t b = bitsOf(x);
    int64_t e = int64_t(b >> 52) - 1023;
    double m = fromBits((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull); // [1, 2)
    if (m > 1.4142135623730951) {
        m = m * 0.5;
        e = e + 1;
    }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 0.034482758620689655;
    p = p * z + 0.037037037037037035;
    p = p * z + 0.04;
    p = p * z + 0.043478260869565216;
    p = p * z + 0.047619047619047616;
    p = p * z + 0.05263157894736842;
    p = p * z + 0.058823529411764705;
    p = p * z + 0.06666666666666667;
    p = p * z + 0.07692307692307693;
    p = p * z + 0.09090909090909091;
    p = p * z + 0.1111111111111111;
    p = p * z + 0.14285714285714285;
    p = p * z + 0.2;
    p = p * z + 0.3333333333333333;
    double lm = 2.0 * (s + s * (z * p));
    double ed = double(e);
    return ed * kLn2Hi + (ed * kLn2Lo + lm);
}

// pow(x, y) for x >= 0 (the only use in the path: clamped cosines and 1 - u);
// x < 0 returns NaN.  Arguments come from floats, so x is never subnormal as a double.
static inline double powd(double x, double y)
{
    if (y == 0.0)
        return 1.0;
    if (x == 1.0)
        return 1.0;
    if (x != x || y != y)
        return x + y;
    if (x < 0.0)
        return (x - x) / (x - x);
    if (x == 0.0)
        return y > 0.0 ? 0.0 : HUGE_VAL;
    if (x == HUGE_VAL)
        return y > 0.0 ? HUGE_VAL : 0.0;
    if (y == HUGE_VAL)
        return x > 1.0 ? HUGE_VAL : 0.0;
    if (y == -HUGE_VAL)
        return x > 1.0 ? 0.0 : HUGE_VAL;
    return expd(y * logd(x));
}

static inline float powf(float x, float y)
{
    return float(powd(double(x), double(y)));
}

// atan(t) for 0 <= t <= 1
static inline double atanUnit(double t)
{
    double base = 0.0;
    if (t > 0.41421356237309503) {
        t = (t - 1.0) / (t + 1.0);
        base = kPio4;
    }
    double z = t * t;
    double p = 0.024390243902439025; // 1/41
    p = p * z + -0.02564102564102564;
    p = p * z + 0.02702702702702703;
    p = p * z + -0.02857142857142857;
    p = p * z + 0.030303030303030304;
    p = p * z + -0.03225806451612903;
    p = p * z + 0.034482758620689655;
    p = p * z + -0.037037037037037035;
    p = p * z + 0.04;
    p = p * z + -0.043478260869565216;
    p = p * z + 0.047619047619047616;
    p = p * z + -0.05263157894736842;
    p = p * z + 0.058823529411764705;
    p = p * z + -0.06666666666666667;
    p = p * z + 0.07692307692307693;
    p = p * z + -0.09090909090909091;
    p = p * z + 0.1111111111111111;
    p = p * z + -0.14285714285714285;
    p = p * z + 0.2;
    p = p * z + -0.3333333333333333;
    return base + (t + t * (z * p));
}

// atan2(y, x) with the C library's conventions for signed zeros; finite arguments
static inline double atan2d(double y, double x)
{
    if (x != x || y != y)
        return x + y;
    bool yneg = std::signbit(y);
    bool xneg = std::signbit(x);
    double ay = yneg ? -y : y;
    double ax = xneg ? -x : x;
    double r;
    if (ay == 0.0 && ax == 0.0) {
        r = 0.0;
    } else if (ay <= ax) {
        r = atanUnit(ay / ax);
    } else {
        r = kPio2 - atanUnit(ax / ay);
    }
    if (xneg)
        r = kPi - r;
    return yneg ? -r : r;
}

static inline float atan2f(float y, float x)
{
    return float(atan2d(double(y), double(x)));
}

// asin(x), |x| <= 1 (callers clamp); outside returns NaN through the square root
static inline float asinf(float xf)
{
    double x = double(xf);
    double c = std::sqrt((1.0 - x) * (1.0 + x));
    return float(atan2d(x, c));
}

static inline float acosf(float xf)
{
    double x = double(xf);
    double s = std::sqrt((1.0 - x) * (1.0 + x));
    return float(atan2d(s, x));
}

} // namespace pmath
