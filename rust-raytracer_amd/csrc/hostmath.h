// Host-side f32 vector / matrix arithmetic with the operation order of cgmath 0.17.0,
// the crate the reference does all its geometry with (Cargo.lock:81-82).  Used by the
// scene builders, the BVH builder and the flattener so that every number uploaded to the
// GPU is the number the reference would have computed.  Compile with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstring>

namespace rtx {

struct V3 {
    float x, y, z;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float magnitude(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalize(V3 a) { return a * (1.0f / magnitude(a)); }

// Column-major 4x4: m[c*4+r].
struct M4 {
    float m[16];
    float &at(int c, int r) { return m[c * 4 + r]; }
    float at(int c, int r) const { return m[c * 4 + r]; }
};
inline M4 m4_identity() {
    M4 r;
    std::memset(r.m, 0, sizeof(r.m));
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
    return r;
}
inline bool m4_equal(const M4 &a, const M4 &b) { return std::memcmp(a.m, b.m, sizeof(a.m)) == 0; }
// Matrix4 * Vector4: ((c0*x + c1*y) + c2*z) + c3*w
inline void m4_mul_v4(const M4 &M, float x, float y, float z, float w, float out[4]) {
    for (int r = 0; r < 4; r++)
        out[r] = ((M.m[0 + r] * x + M.m[4 + r] * y) + M.m[8 + r] * z) + M.m[12 + r] * w;
}
// Matrix4 * Matrix4: column j of the product is lhs * rhs.column(j)
inline M4 operator*(const M4 &a, const M4 &b) {
    M4 r;
    for (int c = 0; c < 4; c++) m4_mul_v4(a, b.m[c * 4], b.m[c * 4 + 1], b.m[c * 4 + 2], b.m[c * 4 + 3], &r.m[c * 4]);
    return r;
}
// Transform::transform_point: Point3::from_homogeneous(M * p.to_homogeneous())
inline V3 transform_point(const M4 &M, V3 p) {
    float h[4];
    m4_mul_v4(M, p.x, p.y, p.z, 1.0f, h);
    float iw = 1.0f / h[3];
    return v3(h[0] * iw, h[1] * iw, h[2] * iw);
}
// Transform::transform_vector: (M * v.extend(0)).truncate()
inline V3 transform_vector(const M4 &M, V3 v) {
    float h[4];
    m4_mul_v4(M, v.x, v.y, v.z, 0.0f, h);
    return v3(h[0], h[1], h[2]);
}

const float kDegToRad = (float)(3.14159265358979323846 / 180.0);

// Matrix4::from_angle_x / _y / _z (angles in degrees, as the scenes use Deg)
inline M4 m4_from_angle_x(float deg) {
    float s = std::sin(deg * kDegToRad), c = std::cos(deg * kDegToRad);
    M4 r = m4_identity();
    r.at(1, 1) = c; r.at(1, 2) = s;
    r.at(2, 1) = -s; r.at(2, 2) = c;
    return r;
}
inline M4 m4_from_angle_y(float deg) {
    float s = std::sin(deg * kDegToRad), c = std::cos(deg * kDegToRad);
    M4 r = m4_identity();
    r.at(0, 0) = c; r.at(0, 2) = -s;
    r.at(2, 0) = s; r.at(2, 2) = c;
    return r;
}
inline M4 m4_from_angle_z(float deg) {
    float s = std::sin(deg * kDegToRad), c = std::cos(deg * kDegToRad);
    M4 r = m4_identity();
    r.at(0, 0) = c; r.at(0, 1) = s;
    r.at(1, 0) = -s; r.at(1, 1) = c;
    return r;
}
inline M4 m4_from_translation(V3 t) {
    M4 r = m4_identity();
    r.at(3, 0) = t.x; r.at(3, 1) = t.y; r.at(3, 2) = t.z;
    return r;
}
inline M4 m4_from_nonuniform_scale(float x, float y, float z) {
    M4 r = m4_identity();
    r.at(0, 0) = x; r.at(1, 1) = y; r.at(2, 2) = z;
    return r;
}
inline float det3(float a0, float a1, float a2, float b0, float b1, float b2, float c0, float c1, float c2) {
    // Matrix3::determinant with columns a,b,c
    return a0 * (b1 * c2 - c1 * b2) - b0 * (a1 * c2 - c1 * a2) + c0 * (a1 * b2 - b1 * a2);
}
// SquareMatrix::invert for Matrix4: cofactors of the transpose scaled by 1/det.
inline bool m4_invert(const M4 &M, M4 &out) {
    // rows of M are the columns of the transpose t
    float t[4][4];  // t[col][row] of transpose = M[row][col]
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) t[c][r] = M.at(r, c);
    auto cof3 = [&](int i, int j) {
        // Matrix3 from the three columns of t other than i, each with element j removed
        float cols[3][3];
        int k = 0;
        for (int c = 0; c < 4; c++) {
            if (c == i) continue;
            int q = 0;
            for (int r = 0; r < 4; r++) {
                if (r == j) continue;
                cols[k][q++] = t[c][r];
            }
            k++;
        }
        return det3(cols[0][0], cols[0][1], cols[0][2], cols[1][0], cols[1][1], cols[1][2], cols[2][0], cols[2][1], cols[2][2]);
    };
    // determinant by expansion along the first row of M: sum_j (-1)^j M[0][j] * minor(0, j)
    float det = 0.0f;
    {
        float d0 = cof3(0, 0), d1 = cof3(0, 1), d2 = cof3(0, 2), d3 = cof3(0, 3);
        det = M.at(0, 0) * d0 - M.at(1, 0) * d1 + M.at(2, 0) * d2 - M.at(3, 0) * d3;
    }
    if (det == 0.0f) return false;
    float inv_det = 1.0f / det;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float sign = ((i + j) & 1) ? -1.0f : 1.0f;
            out.at(i, j) = cof3(i, j) * sign * inv_det;
        }
    return true;
}
// Matrix4::look_at (right-handed): rows (s, u, -f), translation (-eye.s, -eye.u, eye.f)
inline M4 m4_look_at(V3 eye, V3 center, V3 up) {
    V3 f = normalize(center - eye);
    V3 s = normalize(cross(f, up));
    V3 u = cross(s, f);
    M4 r;
    r.at(0, 0) = s.x; r.at(0, 1) = u.x; r.at(0, 2) = -f.x; r.at(0, 3) = 0.0f;
    r.at(1, 0) = s.y; r.at(1, 1) = u.y; r.at(1, 2) = -f.y; r.at(1, 3) = 0.0f;
    r.at(2, 0) = s.z; r.at(2, 1) = u.z; r.at(2, 2) = -f.z; r.at(2, 3) = 0.0f;
    r.at(3, 0) = -dot(eye, s); r.at(3, 1) = -dot(eye, u); r.at(3, 2) = dot(eye, f); r.at(3, 3) = 1.0f;
    return r;
}

}  // namespace rtx
