// posehip_geometry.hpp -- host geometry between the reference's OpenGL scene and the
// PnP stage, header-only C++17 with no dependencies.
//
// The reference renders a 2 x 2.5868 x 0.823 box (src/cereal_box.cpp:98-100) rotating
// about (0.3, 1, 0) (:263) with view = translate(0, 0, -6) (:219) and projection =
// perspective(45 deg, 800/600, 0.1, 100) (:222), reads the frame back with glReadPixels
// (:279, row 0 = BOTTOM of the viewport) and leaves a TODO (:292-295): "map from pixel
// location to 3D point in model coordinates (inverse of projection*view*model)", plus a
// "TEMPORARY" ground-truth model matrix on the wireframe (:286) that the estimated pose
// should replace.  This header is that glue:
//
//   cameraMatrix()            pinhole K equivalent to the GL projection
//   poseFromModelMatrix()     ground-truth (R, t) in OpenCV's camera frame
//   modelMatrixFromPose()     estimated (rvec, tvec) -> GL model matrix for the wireframe
//   liftPixelsToBox()         pixel -> 3D point on the box surface (the TODO)
//
// 4 x 4 matrices are COLUMN-MAJOR arrays of 16, the layout of glm::mat4 (pass
// glm::value_ptr(m)); 3 x 3 matrices and K are row-major arrays of 9, as in posehip.hpp.
// OpenCV's camera frame is x right, y down, z forward; GL eye space is x right, y up,
// z backward: eye = diag(1, -1, -1) * camera.
#ifndef POSEHIP_GEOMETRY_HPP
#define POSEHIP_GEOMETRY_HPP

#include <cmath>
#include <cstddef>
#include <limits>
#include <vector>

namespace posehip {
namespace geometry {

constexpr int SCR_WIDTH = 800, SCR_HEIGHT = 600;                  // src/cereal_box.cpp:30-31
constexpr double BOX_HALF_EXTENTS[3] = {1.0, 1.2934, 0.41149};    // src/cereal_box.cpp:98-100
constexpr double PI = 3.14159265358979323846;

// Rotation vector -> row-major 3 x 3 matrix (cv::Rodrigues, no Jacobian).
inline void rodrigues(const double r[3], double R[9])
{
    const double theta = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < std::numeric_limits<double>::epsilon()) {
        for (int i = 0; i < 9; i++)
            R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    const double k[3] = {r[0] / theta, r[1] / theta, r[2] / theta};
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1 - c;
    const double Kx[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[3 * i + j] = (i == j ? c : 0.0) + c1 * k[i] * k[j] + s * Kx[3 * i + j];
}

// glm::perspective (right-handed, depth -1..1), column-major.
inline void perspective(double fovy_deg, double aspect, double z_near, double z_far, double P[16])
{
    const double f = 1.0 / std::tan(fovy_deg * PI / 180.0 / 2);
    for (int i = 0; i < 16; i++)
        P[i] = 0;
    P[0] = f / aspect;
    P[5] = f;
    P[10] = (z_far + z_near) / (z_near - z_far);
    P[11] = -1.0;
    P[14] = 2 * z_far * z_near / (z_near - z_far);
}

// glm::translate(I, (0, 0, -6)): src/cereal_box.cpp:219.
inline void viewMatrix(double V[16])
{
    for (int i = 0; i < 16; i++)
        V[i] = (i % 5 == 0) ? 1.0 : 0.0;
    V[14] = -6.0;
}

// glm::rotate(I, angle, axis): src/cereal_box.cpp:263 uses axis (0.3, 1, 0).
inline void modelMatrix(double angle_rad, const double axis[3], double M[16])
{
    const double n = std::sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
    const double r[3] = {axis[0] / n * angle_rad, axis[1] / n * angle_rad, axis[2] / n * angle_rad};
    double R[9];
    rodrigues(r, R);
    for (int i = 0; i < 16; i++)
        M[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            M[4 * j + i] = R[3 * i + j];
}

// Pinhole K of the GL projection for pixel rows counted from the TOP: fx = fy =
// (height/2) / tan(fovy/2), principal point at the window centre in continuous window
// coordinates; with pixel_centers the principal point is in OpenCV keypoint coordinates
// (centre of pixel (i, j) at (i, j)): (399.5, 299.5).
inline void cameraMatrix(double K[9], int width = SCR_WIDTH, int height = SCR_HEIGHT, double fovy_deg = 45.0,
                         bool pixel_centers = false)
{
    const double f = (height / 2.0) / std::tan(fovy_deg * PI / 180.0 / 2);
    const double off = pixel_centers ? 0.5 : 0.0;
    const double k[9] = {f, 0, width / 2.0 - off, 0, f, height / 2.0 - off, 0, 0, 1};
    for (int i = 0; i < 9; i++)
        K[i] = k[i];
}

// glReadPixels rows start at the bottom of the viewport: v_from_top = height - 1 - v.
inline double readbackRowToTopDown(double v_from_bottom, int height = SCR_HEIGHT)
{
    return (height - 1) - v_from_bottom;
}

inline void mul4(const double A[16], const double B[16], double C[16]) // C = A * B, column-major
{
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            double s = 0;
            for (int k = 0; k < 4; k++)
                s += A[4 * k + r] * B[4 * c + k];
            C[4 * c + r] = s;
        }
}

// (R, t): model coordinates -> OpenCV camera frame, from the GL model and view matrices.
inline void poseFromModelMatrix(const double model[16], const double view[16], double R[9], double t[3])
{
    double mv[16];
    mul4(view, model, mv);
    const double flip[3] = {1.0, -1.0, -1.0};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            R[3 * i + j] = flip[i] * mv[4 * j + i];
        t[i] = flip[i] * mv[12 + i];
    }
}

// GL model matrix that draws the box at the estimated pose (replaces the ground truth
// handed to the wire shader at src/cereal_box.cpp:286).  view must be a rigid transform
// (rotation + translation), as the reference's is.
inline void modelMatrixFromPose(const double rvec[3], const double tvec[3], const double view[16], double model[16])
{
    double R[9];
    rodrigues(rvec, R);
    const double flip[3] = {1.0, -1.0, -1.0};
    double mv[16] = {0};
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            mv[4 * j + i] = flip[i] * R[3 * i + j];
        mv[12 + i] = flip[i] * tvec[i];
    }
    mv[15] = 1.0;
    double vinv[16] = {0}; // inverse of a rigid transform: [Rv^T | -Rv^T tv]
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            vinv[4 * j + i] = view[4 * i + j];
    for (int i = 0; i < 3; i++)
        vinv[12 + i] = -(vinv[i] * view[12] + vinv[4 + i] * view[13] + vinv[8 + i] * view[14]);
    vinv[15] = 1.0;
    mul4(vinv, mv, model);
}

// The TODO of src/cereal_box.cpp:292-295: model coordinates of the box surface seen at
// each pixel (u, v with rows from the top), given the pose (R, t) of the reference frame.
// pixels: n x 2, points: n x 3 (NaN where the ray misses the box); returns hit flags.
inline std::vector<unsigned char> liftPixelsToBox(const double *pixels, std::size_t n, const double R[9],
                                                  const double t[3], const double K[9], double *points,
                                                  const double half_extents[3] = BOX_HALF_EXTENTS)
{
    std::vector<unsigned char> hit(n, 0);
    double origin[3]; // camera centre in model coordinates: -R^T t
    for (int j = 0; j < 3; j++)
        origin[j] = -(R[j] * t[0] + R[3 + j] * t[1] + R[6 + j] * t[2]);
    const double nan = std::numeric_limits<double>::quiet_NaN();
    const double inf = std::numeric_limits<double>::infinity();
    for (std::size_t i = 0; i < n; i++) {
        const double ray[3] = {(pixels[2 * i] - K[2]) / K[0], (pixels[2 * i + 1] - K[5]) / K[4], 1.0};
        double dir[3];
        for (int j = 0; j < 3; j++)
            dir[j] = R[j] * ray[0] + R[3 + j] * ray[1] + R[6 + j] * ray[2];
        double lo = -inf, hi = inf; // slab method; a NaN bound (0/0) is skipped
        for (int j = 0; j < 3; j++) {
            const double t1 = (-half_extents[j] - origin[j]) / dir[j];
            const double t2 = (half_extents[j] - origin[j]) / dir[j];
            const double a = t1 < t2 ? t1 : t2, b = t1 < t2 ? t2 : t1;
            if (!std::isnan(a) && a > lo)
                lo = a;
            if (!std::isnan(b) && b < hi)
                hi = b;
        }
        const bool ok = lo <= hi && hi > 0;
        hit[i] = ok ? 1 : 0;
        const double s = lo > 0 ? lo : hi;
        for (int j = 0; j < 3; j++)
            points[3 * i + j] = ok ? origin[j] + dir[j] * s : nan;
    }
    return hit;
}

} // namespace geometry
} // namespace posehip

#endif // POSEHIP_GEOMETRY_HPP
