// Stand-in meshes for the four scenes whose .obj files cannot ship with this
// repository (bunny 144 056, erato 412 508, dragon 871 316, aurelius 1 704 768
// faces; include/scene.hpp:294,333,372,411).  A stand-in is a closed, bumpy,
// star-shaped surface (a displaced cube-sphere: evenly shaped triangles like a
// scanned mesh) tessellated to exactly n - 10 triangles, standing on the
// floor of the 10-triangle open box the reference scenes end with
// (tools/range.py:40-54 gives the box faces; :24-32 the ratio of the mesh's
// bounding-box volume to the box volume, 27360/216000).
//
// Everything is a pure function of (preset, n, seed): integer hashing, IEEE
// arithmetic and libm sin/cos, then every coordinate is rounded through the
// "%.9f" text the .obj file carries, so a scene generated in memory equals the
// scene re-loaded from the written file.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "host_scene.hpp"

namespace rtx {

namespace {

inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline double unit(uint64_t h) { return static_cast<double>(h >> 11) * (1.0 / 9007199254740992.0); }

inline double through_text(double v) {
    char buf[64];
    snprintf(buf, sizeof buf, "%.9f", v);
    return strtod(buf, nullptr);
}

}  // namespace

// The second family: a knotted tube.  A closed surface like the first, but concave all over: the
// tube of a (2,3) torus knot passes in front of itself three times, so primary rays see through its
// loops, parts of the model shadow and mirror each other, and reflected rays mostly find geometry
// again -- the ray population of a scanned statue (self-occlusion, self-reflection), which the
// star-shaped blob lacks.  The open box is placed so that, like the reference's own teapot scene
// (camera at three half-sides from the box centre, include/scene.hpp:259), its open front fills the
// view: every primary ray ends on the model or on a wall.
static int generate_knot(const HostScene &probe, uint64_t n_triangles, uint64_t seed, std::vector<double> *verts,
                         std::vector<uint32_t> *faces, std::string *err) {
    const uint64_t target = n_triangles - 10;
    // U segments along the knot, V around the tube: 2 U V triangles; the rest by centroid splits
    uint64_t V = static_cast<uint64_t>(std::sqrt(static_cast<double>(target) / 2.0 / 11.0));
    if (V < 3) V = 3;
    uint64_t U = target / (2 * V);
    if (U < 3) { *err = "knot stand-in needs at least 28 triangles"; return RTX_ERR_INVALID; }
    const uint64_t base = 2 * U * V, extra = (target - base) / 2;
    if (U * V + extra > 0xFFFFFFF0ull) { *err = "stand-in: too many vertices"; return RTX_ERR_INVALID; }
    const double kPi = 3.14159265358979323846;
    const double ph1 = 2 * kPi * unit(mix64(seed ^ 0x41)), ph2 = 2 * kPi * unit(mix64(seed ^ 0x52)),
                 ph3 = 2 * kPi * unit(mix64(seed ^ 0x63));
    std::vector<double> raw;
    raw.reserve(3 * (U * V + extra));
    for (uint64_t i = 0; i < U; i++) {
        const double u = 2 * kPi * static_cast<double>(i) / static_cast<double>(U);
        // centre line C(u) of the (2,3) torus knot and its derivative
        const double c2 = std::cos(2 * u), s2 = std::sin(2 * u), c3 = std::cos(3 * u), s3 = std::sin(3 * u);
        const double C[3] = {(2 + c3) * c2, (2 + c3) * s2, s3};
        double T[3] = {-3 * s3 * c2 - 2 * (2 + c3) * s2, -3 * s3 * s2 + 2 * (2 + c3) * c2, 3 * c3};
        const double tl = std::sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2]);
        for (double &t : T) t /= tl;
        // frame: B = T x C (C is never parallel to T on this curve), N = B x T
        double B[3] = {T[1] * C[2] - T[2] * C[1], T[2] * C[0] - T[0] * C[2], T[0] * C[1] - T[1] * C[0]};
        const double bl = std::sqrt(B[0] * B[0] + B[1] * B[1] + B[2] * B[2]);
        for (double &b : B) b /= bl;
        const double N[3] = {B[1] * T[2] - B[2] * T[1], B[2] * T[0] - B[0] * T[2], B[0] * T[1] - B[1] * T[0]};
        for (uint64_t j = 0; j < V; j++) {
            const double v = 2 * kPi * static_cast<double>(j) / static_cast<double>(V);
            double rho = 0.46;
            rho *= 1.0 + 0.16 * std::sin(5 * u + ph1) * std::cos(3 * v + ph2) + 0.07 * std::sin(17 * u + ph3) * std::sin(7 * v + ph1) +
                   0.025 * std::sin(61 * u + ph2) * std::cos(13 * v + ph3) +
                   0.012 * (unit(mix64(seed * 0x100000001B3ull + i * 1048576ull + j)) - 0.5);        // scan-like roughness
            const double cv = std::cos(v), sv = std::sin(v);
            for (int k = 0; k < 3; k++) raw.push_back(C[k] + rho * (cv * N[k] + sv * B[k]));
        }
    }
    faces->clear();
    faces->reserve(3 * n_triangles);
    auto tri = [&](uint64_t a, uint64_t b, uint64_t c) {
        faces->push_back(static_cast<uint32_t>(a)); faces->push_back(static_cast<uint32_t>(b)); faces->push_back(static_cast<uint32_t>(c));
    };
    for (uint64_t i = 0; i < U; i++)
        for (uint64_t j = 0; j < V; j++) {
            const uint64_t i1 = (i + 1) % U, j1 = (j + 1) % V;
            const uint64_t a = i * V + j, av = i * V + j1, au = i1 * V + j, auv = i1 * V + j1;
            tri(a, av, au);        // (around the tube) x (along it) points away from the centre line
            tri(auv, au, av);
        }
    uint64_t next_vertex = raw.size() / 3;
    for (uint64_t e = 0; e < extra; e++) {
        const uint64_t f = (e * base) / extra;
        const uint32_t a = (*faces)[3 * f], b = (*faces)[3 * f + 1], c = (*faces)[3 * f + 2];
        for (int k = 0; k < 3; k++) raw.push_back((raw[3 * a + k] + raw[3 * b + k] + raw[3 * c + k]) / 3);
        const uint64_t v = next_vertex++;
        (*faces)[3 * f + 2] = static_cast<uint32_t>(v);
        tri(b, c, v);
        tri(c, a, v);
    }
    // The box: a cube of side 2 * camera height on the floor y = 0, centred under the camera in x and
    // 1.5 sides in front of it in z (its open front then spans the viewport exactly); the knot's
    // bounding box is fitted to 0.56 x 0.56 x 0.40 of the side, standing on the floor, in the middle.
    const double side = 2 * probe.camera_position[1], cx = probe.camera_position[0], cz = probe.camera_position[2] + 1.5 * side;
    const double half_w = 0.28 * side, height = 0.56 * side, half_d = 0.20 * side;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    const uint64_t nv_model = raw.size() / 3;
    for (uint64_t v = 0; v < nv_model; v++)
        for (int k = 0; k < 3; k++) {
            if (raw[3 * v + k] < lo[k]) lo[k] = raw[3 * v + k];
            if (raw[3 * v + k] > hi[k]) hi[k] = raw[3 * v + k];
        }
    verts->clear();
    verts->reserve(3 * (nv_model + 8));
    for (uint64_t v = 0; v < nv_model; v++) {
        const double x = cx + (2 * (raw[3 * v] - lo[0]) / (hi[0] - lo[0]) - 1) * half_w;
        const double y = 0.04 * side + (raw[3 * v + 1] - lo[1]) / (hi[1] - lo[1]) * height;
        const double z = cz + (2 * (raw[3 * v + 2] - lo[2]) / (hi[2] - lo[2]) - 1) * half_d;
        verts->push_back(through_text(x)); verts->push_back(through_text(y)); verts->push_back(through_text(z));
    }
    const double h = side / 2;
    const double bx[8][3] = {{cx - h, 0, cz - h}, {cx + h, 0, cz - h}, {cx + h, 0, cz + h}, {cx - h, 0, cz + h},
                             {cx - h, side, cz - h}, {cx + h, side, cz - h}, {cx + h, side, cz + h}, {cx - h, side, cz + h}};
    for (const auto &p : bx) for (int k = 0; k < 3; k++) verts->push_back(through_text(p[k]));
    static const int box_faces[10][3] = {{1, 2, 3}, {3, 4, 1}, {5, 6, 7}, {7, 8, 5}, {4, 3, 7},
                                         {7, 8, 4}, {3, 2, 6}, {6, 7, 3}, {1, 4, 8}, {8, 5, 1}};       // tools/range.py:40-51
    for (const auto &f : box_faces) tri(nv_model + f[0] - 1, nv_model + f[1] - 1, nv_model + f[2] - 1);
    if (faces->size() / 3 != n_triangles) { *err = "stand-in: internal total mismatch"; return RTX_ERR_INVALID; }
    return RTX_OK;
}

int generate_standin(const std::string &preset, uint64_t n_triangles, uint64_t seed, int family,
                     std::vector<double> *verts, std::vector<uint32_t> *faces, std::string *err) {
    HostScene probe;
    probe.vertices.assign(90, 0.0);
    if (int rc = apply_preset(&probe, preset, err)) return rc;
    if (n_triangles < 10 + 12) { *err = "stand-in needs at least 22 triangles"; return RTX_ERR_INVALID; }
    if ((n_triangles - 10) % 2) { *err = "stand-in triangle count must be even"; return RTX_ERR_INVALID; }
    if (family == 1) return n_triangles < 10 + 28 ? (*err = "knot stand-in needs at least 38 triangles", RTX_ERR_INVALID)
                                                  : generate_knot(probe, n_triangles, seed, verts, faces, err);
    if (family != 0) { *err = "unknown stand-in family"; return RTX_ERR_INVALID; }
    const uint64_t target = n_triangles - 10;

    // Cube-sphere: each cube face carries an n x n grid of quads (two triangles each),
    // pushed out to a bumpy star-shaped surface -> 12 n^2 well-shaped triangles and no
    // pole fans; the remainder is made up by splitting triangles at their centroid (+2 each).
    uint64_t n = static_cast<uint64_t>(std::sqrt(static_cast<double>(target) / 12.0));
    if (n < 1) n = 1;
    while (12 * n * n > target) n--;
    if (n < 1) { *err = "stand-in needs at least 22 triangles"; return RTX_ERR_INVALID; }
    const uint64_t base = 12 * n * n;
    const uint64_t extra = (target - base) / 2;
    if (extra > base) { *err = "stand-in: unsupported triangle count"; return RTX_ERR_INVALID; }

    // box: side = 2 * camera height, centred under the camera in x, around z = 0
    const double side = 2 * probe.camera_position[1];
    const double cx = probe.camera_position[0];
    // mesh bounding box as a fraction of the box side: 0.50 * 0.62 * 0.41 = 0.1271
    const double half_w = 0.25 * side, height = 0.62 * side, half_d = 0.205 * side;

    const double kPi = 3.14159265358979323846;
    const double ph1 = 2 * kPi * unit(mix64(seed ^ 0x11)), ph2 = 2 * kPi * unit(mix64(seed ^ 0x22)),
                 ph3 = 2 * kPi * unit(mix64(seed ^ 0x33));
    // A point of the cube surface has integer coordinates (a,b,c) in [0,n]^3 with one of
    // them 0 or n; everything below is a function of that triple only, so the copies of a
    // point that two or three faces share come out bit-identical (a closed surface).
    auto surface_point = [&](uint64_t a, uint64_t b, uint64_t c, double out[3]) {
        const double x = -1.0 + 2.0 * static_cast<double>(a) / static_cast<double>(n);
        const double y = -1.0 + 2.0 * static_cast<double>(b) / static_cast<double>(n);
        const double z = -1.0 + 2.0 * static_cast<double>(c) / static_cast<double>(n);
        const double len = std::sqrt(x * x + y * y + z * z);
        const double dx = x / len, dy = y / len, dz = z / len;
        double r = 1.0;
        r += 0.22 * std::sin(3.0 * dy + ph1) * std::cos(2.5 * dx + ph2);
        r += 0.10 * std::sin(8.0 * dx + ph2) * std::sin(7.0 * dz + ph3);
        r += 0.035 * std::sin(27.0 * dy + ph3) * std::cos(23.0 * dz + ph1) * std::sin(19.0 * dx);
        r += 0.010 * (unit(mix64(seed * 0x100000001B3ull + ((a * 2097152ull + b) * 2097152ull + c))) - 0.5);   // scan-like roughness
        out[0] = r * dx; out[1] = r * dy; out[2] = r * dz;
    };

    std::vector<double> raw;   // model vertices before fitting
    raw.reserve(3 * (6 * (n + 1) * (n + 1) + extra));
    auto push = [&](double x, double y, double z) { raw.push_back(x); raw.push_back(y); raw.push_back(z); };
    faces->clear();
    faces->reserve(3 * n_triangles);
    auto tri = [&](uint64_t a, uint64_t b, uint64_t c) {
        faces->push_back(static_cast<uint32_t>(a)); faces->push_back(static_cast<uint32_t>(b)); faces->push_back(static_cast<uint32_t>(c));
    };
    // face f: fixed axis f/2 at 0 (f even) or n (f odd); (i,j) run over the other two axes
    for (int f = 0; f < 6; f++) {
        const int axis = f / 2;
        const uint64_t fixed = (f & 1) ? n : 0;
        const uint64_t v0 = raw.size() / 3;
        for (uint64_t j = 0; j <= n; j++)
            for (uint64_t i = 0; i <= n; i++) {
                uint64_t abc[3];
                abc[axis] = fixed;
                abc[(axis + 1) % 3] = i;
                abc[(axis + 2) % 3] = j;
                double p[3];
                surface_point(abc[0], abc[1], abc[2], p);
                push(p[0], p[1], p[2]);
            }
        for (uint64_t j = 0; j < n; j++)
            for (uint64_t i = 0; i < n; i++) {
                const uint64_t a = v0 + j * (n + 1) + i, b = a + 1, c = a + (n + 1), d = c + 1;
                if (f & 1) { tri(a, b, d); tri(d, c, a); }      // outward winding on both sides of the axis
                else       { tri(a, c, d); tri(d, b, a); }
            }
    }
    if (faces->size() / 3 != base) { *err = "stand-in: internal face count mismatch"; return RTX_ERR_INVALID; }

    // centroid splits, spread evenly over the base triangles
    uint64_t next_vertex = raw.size() / 3;
    for (uint64_t e = 0; e < extra; e++) {
        const uint64_t f = (e * base) / extra;
        uint32_t a = (*faces)[3 * f], b = (*faces)[3 * f + 1], c = (*faces)[3 * f + 2];
        double m[3];
        for (int k = 0; k < 3; k++) m[k] = (raw[3 * a + k] + raw[3 * b + k] + raw[3 * c + k]) / 3;
        push(m[0] * 1.002, m[1] * 1.002, m[2] * 1.002);
        const uint64_t v = next_vertex++;
        (*faces)[3 * f + 2] = static_cast<uint32_t>(v);       // (a, b, v)
        tri(b, c, v);
        tri(c, a, v);
    }

    // fit the model's bounding box to half_w x height x half_d, standing on y = 0
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    const uint64_t nv_model = raw.size() / 3;
    for (uint64_t v = 0; v < nv_model; v++)
        for (int k = 0; k < 3; k++) {
            if (raw[3 * v + k] < lo[k]) lo[k] = raw[3 * v + k];
            if (raw[3 * v + k] > hi[k]) hi[k] = raw[3 * v + k];
        }
    verts->clear();
    verts->reserve(3 * (nv_model + 8));
    for (uint64_t v = 0; v < nv_model; v++) {
        double x = cx + (2 * (raw[3 * v] - lo[0]) / (hi[0] - lo[0]) - 1) * half_w;
        double y = (raw[3 * v + 1] - lo[1]) / (hi[1] - lo[1]) * height;
        double z = (2 * (raw[3 * v + 2] - lo[2]) / (hi[2] - lo[2]) - 1) * half_d;
        verts->push_back(through_text(x)); verts->push_back(through_text(y)); verts->push_back(through_text(z));
    }
    // the box: floor y = 0, ceiling y = side, back wall z = +side/2, side walls; the
    // front (towards the camera) stays open -- vertex order as teapot.obj's last 8
    const double h = side / 2;
    const double bx[8][3] = {{cx - h, 0, -h}, {cx + h, 0, -h}, {cx + h, 0, h}, {cx - h, 0, h},
                             {cx - h, side, -h}, {cx + h, side, -h}, {cx + h, side, h}, {cx - h, side, h}};
    for (const auto &p : bx) for (int k = 0; k < 3; k++) verts->push_back(through_text(p[k]));
    // tools/range.py:40-51 (1-based there)
    static const int box_faces[10][3] = {{1, 2, 3}, {3, 4, 1}, {5, 6, 7}, {7, 8, 5}, {4, 3, 7},
                                         {7, 8, 4}, {3, 2, 6}, {6, 7, 3}, {1, 4, 8}, {8, 5, 1}};
    for (const auto &f : box_faces) tri(nv_model + f[0] - 1, nv_model + f[1] - 1, nv_model + f[2] - 1);
    if (faces->size() / 3 != n_triangles) { *err = "stand-in: internal total mismatch"; return RTX_ERR_INVALID; }
    return RTX_OK;
}

// The reference's OBJ dialect: "v x y z" and "f i j k" (1-based), nothing else,
// and NO trailing newline (see parse_obj for why that matters).
int write_obj(const std::string &path, const std::vector<double> &verts, const std::vector<uint32_t> &faces,
              std::string *err) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { *err = "Failed to open file.\n"; return RTX_ERR_IO; }
    std::vector<char> buf(1 << 20);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    for (size_t v = 0; v + 2 < verts.size(); v += 3) fprintf(f, "v %.9f %.9f %.9f\n", verts[v], verts[v + 1], verts[v + 2]);
    const size_t nf = faces.size() / 3;
    for (size_t t = 0; t < nf; t++)
        fprintf(f, t + 1 < nf ? "f %u %u %u\n" : "f %u %u %u", faces[3 * t] + 1, faces[3 * t + 1] + 1, faces[3 * t + 2] + 1);
    int bad = ferror(f);
    bad |= fclose(f);
    if (bad) { *err = "write failed: " + path; return RTX_ERR_IO; }
    return RTX_OK;
}

}  // namespace rtx
