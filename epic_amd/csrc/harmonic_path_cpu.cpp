// harmonic_path_cpu.cpp -- streamline extraction on the converged field (host side), single and double precision.
//
// Consumers of the relaxation's result: bilinear potential, normalised central-difference gradient, "stuck" detector
// and the gradient-ascent streamline.  Restates libepic/src/harmonic/harmonic_path_cpu.cpp:41-232 (float, log-space
// field inside a Harmonic) and libepic/src/harmonic/harmonic_legacy_path_cpu.cpp (double, linear-space field passed as
// w/h/locked/u, ascent or descent by `flipped`) as ONE template, since the two reference files differ only in the
// scalar type, the validity rule of the start cell, the step direction and the length bound.  O(path length) host
// work; it is exported because the ROS callers link it (src/epic_nav_core_plugin.cpp:291-298,
// src/epic_navigation_node_harmonic.cpp:614-674) and free the result with delete[] -- so it must come from new[].
//
// The arithmetic keeps the reference's types step by step (including its std::pow(x, 2) on a float, which C++11
// evaluates in double), so results are bit-identical: tests/test_path_cpu.py.
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <vector>

#include "../../include/epic/epic_abi.h"

namespace {

constexpr unsigned kStuckHistory = 5;  // PATH_STUCK_HISTORY_LENGTH, harmonic_path_cpu.cpp:39

template <typename T>
struct Field {
    unsigned w, h;           // columns, rows
    const unsigned *locked;
    const T *u;
    T at(unsigned x, unsigned y) const { return u[(size_t)y * w + x]; }
    unsigned lock(unsigned x, unsigned y) const { return locked[(size_t)y * w + x]; }
};

// How a start / sample cell is judged unusable.  Log-space: a locked cell with u < 0 is an obstacle (goals are
// locked with u == 0), harmonic_path_cpu.cpp:55-57.  Legacy potential: same test on the linear field,
// harmonic_legacy_path_cpu.cpp:52-54.  Legacy path start: obstacle value depends on `flipped`, :163-166.
enum class Rule { kNegativeIsObstacle, kLegacyStart };

template <typename T>
bool usable(const Field<T> &f, T x, T y, Rule rule, int flipped)
{
    const unsigned cx = (unsigned)(x + T(0.5)), cy = (unsigned)(y + T(0.5));
    if (cx >= f.w || cy >= f.h) return false;
    if (f.lock(cx, cy) != 1) return true;
    if (rule == Rule::kNegativeIsObstacle) return !(f.at(cx, cy) < T(0));
    return !((flipped == 0 && f.at(cx, cy) == T(1)) || (flipped == 1 && f.at(cx, cy) == T(0)));
}

// Bilinear interpolation between the four cell centres around (x, y): harmonic_path_cpu.cpp:63-80.
template <typename T>
T bilinear(const Field<T> &f, T x, T y)
{
    const unsigned x0 = (unsigned)(x - T(0.5)), x1 = (unsigned)(x + T(0.5));
    const unsigned y0 = (unsigned)(y - T(0.5)), y1 = (unsigned)(y + T(0.5));
    const T alpha = x - x0, beta = y - y0;
    const T top = (T(1) - alpha) * f.at(x0, y0) + alpha * f.at(x1, y0);
    const T bottom = (T(1) - alpha) * f.at(x0, y1) + alpha * f.at(x1, y1);
    return (T(1) - beta) * top + beta * bottom;
}

template <typename T>
int potential(const Field<T> &f, T x, T y, T &out, const char *fn)
{
    if (!usable(f, x, y, Rule::kNegativeIsObstacle, 0)) {
        fprintf(stderr, "Error[%s]: %s\n", fn, "Invalid location.");
        return EPIC_ERROR_INVALID_LOCATION;
    }
    out = bilinear(f, x, y);
    return EPIC_SUCCESS;
}

// Central differences at +-cd, normalised to unit length: harmonic_path_cpu.cpp:85-118.  The norm is evaluated in
// double even for float fields (std::pow(float, int) promotes), then narrowed.
template <typename T>
int gradient(const Field<T> &f, T x, T y, T cd, T &gx, T &gy, const char *fn, const char *fn_potential)
{
    T v[4] = {T(0), T(0), T(0), T(0)};
    int rc = potential(f, x - cd, y, v[0], fn_potential);
    rc += potential(f, x + cd, y, v[1], fn_potential);
    rc += potential(f, x, y - cd, v[2], fn_potential);
    rc += potential(f, x, y + cd, v[3], fn_potential);
    if (rc != EPIC_SUCCESS) {
        fprintf(stderr, "Error[%s]: %s\n", fn, "Failed to compute potential values.");
        return EPIC_ERROR_INVALID_GRADIENT;
    }
    gx = (v[1] - v[0]) / (T(2) * cd);
    gy = (v[3] - v[2]) / (T(2) * cd);
    const T norm = (T)std::sqrt(std::pow(gx, 2) + std::pow(gy, 2));  // same overloads as the reference: double
    gx /= norm;
    gy /= norm;
    return EPIC_SUCCESS;
}

// True when the newest point came back to within stepSize/2 of one of the previous kStuckHistory points
// (harmonic_path_cpu.cpp:121-151).  An odd-length vector counts as stuck, an empty one as fine.
template <typename T>
bool stuck(const std::vector<T> &p, T stepSize)
{
    const unsigned n = (unsigned)p.size();
    if (n % 2 == 1) return true;
    if (n == 0) return false;
    const T x = p[n - 2], y = p[n - 1];
    const int floor_i = std::max(0, (int)n - 2 * (int)kStuckHistory - 2);
    for (unsigned i = n - 2; i > (unsigned)floor_i; i -= 2) {
        const T dx = x - p[i - 2], dy = y - p[i - 1];
        const T dist = (T)std::sqrt(std::pow(dx, 2) + std::pow(dy, 2));
        if (dist < stepSize / T(2)) return true;
    }
    return false;
}

struct Names {
    const char *path, *gradient, *potential;
};

// Follow the gradient from (x, y) until a locked cell is entered, the walk gets stuck, or the length bound is hit:
// harmonic_path_cpu.cpp:154-221 / harmonic_legacy_path_cpu.cpp:152-222.
template <typename T>
int follow(const Field<T> &f, T x, T y, T stepSize, T cd, size_t max_values, Rule start_rule, int flipped, T sign,
           unsigned &k, T *&path, const Names &nm)
{
    if (!usable(f, x, y, start_rule, flipped)) {
        fprintf(stderr, "Error[%s]: %s\n", nm.path, "Invalid location.");
        return EPIC_ERROR_INVALID_LOCATION;
    }
    std::vector<T> pts;
    pts.push_back(x);
    pts.push_back(y);
    unsigned cx = (unsigned)(x + T(0.5)), cy = (unsigned)(y + T(0.5));
    while (f.lock(cx, cy) != 1 && !stuck(pts, stepSize) && pts.size() < max_values) {
        T gx = T(0), gy = T(0);
        if (gradient(f, x, y, cd, gx, gy, nm.gradient, nm.potential) != EPIC_SUCCESS) {
            fprintf(stderr, "Error[%s]: %s\n", nm.path, "Could not compute gradient.");
            return EPIC_ERROR_INVALID_GRADIENT;
        }
        if (sign > T(0)) {
            x += gx * stepSize;
            y += gy * stepSize;
        } else {
            x -= gx * stepSize;
            y -= gy * stepSize;
        }
        pts.push_back(x);
        pts.push_back(y);
        cx = (unsigned)(x + T(0.5));
        cy = (unsigned)(y + T(0.5));
    }
    // two points or fewer: the gradient was degenerate, i.e. the field is not relaxed enough here
    if (pts.size() / 2 <= 2) {
        fprintf(stderr, "Error[%s]: %s\n", nm.path, "Could not compute a valid path.");
        return EPIC_ERROR_INVALID_PATH;
    }
    k = (unsigned)(pts.size() / 2);
    path = new T[2 * (size_t)k];  // callers release it with delete[] (epic_nav_core_plugin.cpp:303-305, :330)
    for (size_t i = 0; i < 2 * (size_t)k; i++) path[i] = pts[i];
    return EPIC_SUCCESS;
}

bool valid(const epic::Harmonic *h) { return h && h->m && h->u && h->locked; }

Field<float> field_of(const epic::Harmonic *h) { return Field<float>{h->m[1], h->m[0], h->locked, h->u}; }

const Names kNames = {"harmonic_compute_path_2d_cpu", "harmonic_compute_gradient_2d_cpu",
                      "harmonic_compute_potential_2d_cpu"};
const Names kLegacyNames = {"harmonic_legacy_compute_path_2d_cpu", "harmonic_legacy_compute_gradient_2d_cpu",
                            "harmonic_legacy_compute_potential_2d_cpu"};

}  // namespace

namespace epic {
extern "C" {

int harmonic_compute_potential_2d_cpu(Harmonic *harmonic, float x, float y, float &potential_out)
{
    if (!valid(harmonic)) {
        fprintf(stderr, "Error[%s]: %s\n", kNames.potential, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    return potential(field_of(harmonic), x, y, potential_out, kNames.potential);
}

int harmonic_compute_gradient_2d_cpu(Harmonic *harmonic, float x, float y, float cdPrecision, float &partialX,
                                     float &partialY)
{
    if (!valid(harmonic)) {
        fprintf(stderr, "Error[%s]: %s\n", kNames.gradient, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    return gradient(field_of(harmonic), x, y, cdPrecision, partialX, partialY, kNames.gradient, kNames.potential);
}

int harmonic_compute_path_2d_cpu(Harmonic *harmonic, float x, float y, float stepSize, float cdPrecision,
                                 unsigned int maxLength, unsigned int &k, float *&path)
{
    if (!valid(harmonic) || path != nullptr) {
        fprintf(stderr, "Error[%s]: %s\n", kNames.path, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    // the reference compares size() with the unsigned product 2 * maxLength (harmonic_path_cpu.cpp:185)
    return follow<float>(field_of(harmonic), x, y, stepSize, cdPrecision, (size_t)(2u * maxLength),
                         Rule::kNegativeIsObstacle, 0, 1.0f, k, path, kNames);
}

int harmonic_free_path_cpu(float *&path)
{
    delete[] path;
    path = nullptr;
    return EPIC_SUCCESS;
}

int harmonic_legacy_compute_potential_2d_cpu(unsigned int w, unsigned int h, unsigned int *locked, double *u, double x,
                                             double y, double &potential_out)
{
    if (w == 0 || h == 0 || locked == nullptr || u == nullptr) {
        fprintf(stderr, "Error[%s]: %s\n", kLegacyNames.potential, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    return potential(Field<double>{w, h, locked, u}, x, y, potential_out, kLegacyNames.potential);
}

int harmonic_legacy_compute_gradient_2d_cpu(unsigned int w, unsigned int h, unsigned int *locked, double *u, double x,
                                            double y, double cdPrecision, double &partialX, double &partialY)
{
    if (w == 0 || h == 0 || locked == nullptr || u == nullptr) {
        fprintf(stderr, "Error[%s]: %s\n", kLegacyNames.gradient, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    return gradient(Field<double>{w, h, locked, u}, x, y, cdPrecision, partialX, partialY, kLegacyNames.gradient,
                    kLegacyNames.potential);
}

int harmonic_legacy_compute_path_2d_cpu(unsigned int w, unsigned int h, unsigned int *locked, double *u, double x,
                                        double y, double stepSize, double cdPrecision, unsigned int maxLength,
                                        int flipped, unsigned int &k, double *&path)
{
    if (w == 0 || h == 0 || locked == nullptr || u == nullptr || path != nullptr) {
        fprintf(stderr, "Error[%s]: %s\n", kLegacyNames.path, "Invalid data.");
        return EPIC_ERROR_INVALID_DATA;
    }
    // legacy bound: size() < maxLength (values, not points); step = +gradient when flipped == 1, -gradient otherwise
    return follow<double>(Field<double>{w, h, locked, u}, x, y, stepSize, cdPrecision, (size_t)maxLength,
                          Rule::kLegacyStart, flipped, flipped == 1 ? 1.0 : -1.0, k, path, kLegacyNames);
}

int harmonic_legacy_free_path_cpu(double *&path)
{
    delete[] path;
    path = nullptr;
    return EPIC_SUCCESS;
}

}  // extern "C"
}  // namespace epic
