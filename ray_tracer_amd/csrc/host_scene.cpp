// Host scene: OBJ loader, scene factories, triangle setup and the BVH builder.
// Citations are file:line in khaganmv/ray-tracer.
#include "host_scene.hpp"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace rtx {

namespace {

const double kInf = std::numeric_limits<double>::infinity();

// The whole file, NUL-terminated (strtod may look one character past the last number).  Mapped
// when the size leaves room for the terminator in the last page (the kernel zero-fills it),
// read into a buffer otherwise.
struct FileText {
    const char *data = nullptr;
    size_t size = 0;
    void *mapping = nullptr;
    size_t mapped = 0;
    std::unique_ptr<char[]> buffer;
    FileText() = default;
    FileText(const FileText &) = delete;
    FileText &operator=(const FileText &) = delete;
    ~FileText() { if (mapping) munmap(mapping, mapped); }
};

bool slurp(const std::string &path, FileText *out) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 0) { close(fd); return false; }
    out->size = static_cast<size_t>(st.st_size);
    const size_t page = static_cast<size_t>(sysconf(_SC_PAGESIZE));
    if (out->size && out->size % page != 0) {
        void *m = mmap(nullptr, out->size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        if (m != MAP_FAILED) {
            out->mapping = m;
            out->mapped = out->size;
            out->data = static_cast<const char *>(m);
            close(fd);
            return true;
        }
    }
    out->buffer.reset(new char[out->size + 1]);
    size_t got = 0;
    while (got < out->size) {
        const ssize_t k = read(fd, out->buffer.get() + got, out->size - got);
        if (k <= 0) break;
        got += static_cast<size_t>(k);
    }
    close(fd);
    out->buffer[out->size] = 0;
    out->data = out->buffer.get();
    return got == out->size;
}

// strtod for the decimal numbers OBJ files hold.  A literal of at most 19 significant digits whose
// digit string w fits 2^53 and whose decimal exponent e is within +-22 is w * 10^e with both factors
// exact doubles, so one IEEE multiplication or division rounds it correctly: the same bits as
// strtod (and as the reference's `fs >> double`).  Everything else (longer literals, large
// exponents, inf/nan, hex) goes to strtod itself.  Leading white space is skipped like strtod does.
double scan_double(const char *p, char **end_out) {
    static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                      1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char *q = p;
    while (*q == ' ' || (*q >= '\t' && *q <= '\r')) q++;
    bool negative = false;
    if (*q == '-' || *q == '+') negative = *q++ == '-';
    uint64_t w = 0;
    int digits = 0, exp10 = 0;
    bool any = false;
    for (; *q >= '0' && *q <= '9'; q++) {
        any = true;
        if (w || *q != '0') { w = w * 10 + static_cast<uint64_t>(*q - '0'); digits++; }
    }
    if (*q == '.') {
        q++;
        for (; *q >= '0' && *q <= '9'; q++) {
            any = true;
            if (w || *q != '0') { w = w * 10 + static_cast<uint64_t>(*q - '0'); digits++; }
            exp10--;
        }
    }
    if (!any || digits > 19 || *q == 'x' || *q == 'X') return strtod(p, end_out);
    if (*q == 'e' || *q == 'E') {
        const char *r = q + 1;
        bool eneg = false;
        if (*r == '-' || *r == '+') eneg = *r++ == '-';
        if (*r >= '0' && *r <= '9') {
            int e = 0;
            for (; *r >= '0' && *r <= '9'; r++)
                if (e < 10000) e = e * 10 + (*r - '0');
            exp10 += eneg ? -e : e;
            q = r;
        }
    }
    if (w > (1ull << 53) || exp10 < -22 || exp10 > 22) return strtod(p, end_out);
    double v = static_cast<double>(w);
    v = exp10 < 0 ? v / kPow10[-exp10] : v * kPow10[exp10];
    *end_out = const_cast<char *>(q);
    return negative ? -v : v;
}

// strtoll(p, &end, 10) for literals of up to 18 digits; longer ones go to strtoll.
long long scan_integer(const char *p, char **end_out) {
    const char *q = p;
    while (*q == ' ' || (*q >= '\t' && *q <= '\r')) q++;
    bool negative = false;
    if (*q == '-' || *q == '+') negative = *q++ == '-';
    const char *first = q;
    long long v = 0;
    for (; *q >= '0' && *q <= '9'; q++) v = v * 10 + (*q - '0');
    if (q == first || q - first > 18) return strtoll(p, end_out, 10);
    *end_out = const_cast<char *>(q);
    return negative ? -v : v;
}

// RTX_TIMING=1 prints how long each stage of a loader call took (stderr).
struct StageClock {
    const bool on = getenv("RTX_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[rtx timing] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// Threads for the host-side loops: OMP_NUM_THREADS when set (GPU boxes export their CPU share
// there), else the hardware's count, never more than 16.
size_t worker_threads() {
    size_t n = std::thread::hardware_concurrency();
    if (const char *v = getenv("OMP_NUM_THREADS")) {
        const long k = atol(v);
        if (k > 0) n = static_cast<size_t>(k);
    }
    return std::max<size_t>(1, std::min<size_t>(n, 16));
}

template <typename F>
void parallel_for(size_t n_items, F fn) {
    const size_t n_threads = std::min(worker_threads(), n_items);
    if (n_threads <= 1) {
        for (size_t i = 0; i < n_items; i++) fn(i);
        return;
    }
    std::vector<std::thread> pool;
    std::atomic<size_t> next{0};
    for (size_t t = 0; t < n_threads; t++)
        pool.emplace_back([&]() { for (size_t i = next++; i < n_items; i = next++) fn(i); });
    for (std::thread &t : pool) t.join();
}

}  // namespace

void parallel_copy(void *dst, const void *src, size_t bytes) {
    const size_t piece = 1u << 20;
    if (bytes < 4 * piece) { memcpy(dst, src, bytes); return; }
    parallel_for((bytes + piece - 1) / piece, [&](size_t k) {
        const size_t at = k * piece;
        memcpy(static_cast<char *>(dst) + at, static_cast<const char *>(src) + at, std::min(piece, bytes - at));
    });
}

// Scene::parseOBJ, include/scene.hpp:213-253.
//
// The reference tokenises with `fs >> id` (one non-blank char), then `fs >> x >> y >> z`
// for 'v' or `fs >> i >> j >> k` for 'f'; any other char is dropped.  Its loop
// condition is `!fs.eof()`, so a file that ends in white space runs the body once
// more with every extraction failing and the locals still holding the previous
// record: the last record is pushed twice (teapot.obj plus a newline loads 6331
// triangles with the reference as built from source).  Kept here, because the scene
// factories address materials relative to the END of the triangle list.
// A 'v' or 'f' followed by something that is not three numbers leaves the
// reference's stream failed and its loop spinning; here that is an error.
//
// The text is cut into pieces at line starts that open a record ("\nv" / "\nf") and the
// pieces are tokenised on separate threads with the same grammar; a piece boundary is never
// inside a record of a file the reference can read (a record's three numbers are followed by
// the next record's letter).  Faces keep their 1-based indices until every piece's vertex
// count is known.
namespace {

struct ObjPiece {
    std::vector<double> pos;          // 3 per vertex
    std::vector<long long> faces;     // i j k, then the number of vertices defined before the face in this piece
    std::string err;
    int rc = RTX_OK;
};

void parse_piece(const char *p, const char *const end, bool last_piece, ObjPiece *out) {
    char id = 0;
    double xyz[3] = {0, 0, 0};
    long long ijk[3] = {0, 0, 0};
    for (bool done = false; !done;) {
        while (p < end && isspace(static_cast<unsigned char>(*p))) p++;
        const bool fresh = p < end;        // false: ran into the end, locals are stale
        if (fresh) id = *p++;
        else if (last_piece) done = true;  // end-of-file: the reference's loop body runs once more
        else break;
        if (id == 'v') {
            if (fresh) {
                for (double &c : xyz) {
                    char *q = nullptr;
                    c = scan_double(p, &q);
                    if (q == p || q > end) { out->err = "OBJ: vertex record is not 'v x y z'"; out->rc = RTX_ERR_UNSUPPORTED; return; }
                    p = q;
                }
                if (last_piece && p >= end) done = true;   // last number touched end-of-file: eofbit set
            }
            out->pos.insert(out->pos.end(), xyz, xyz + 3);
        } else if (id == 'f') {
            if (fresh) {
                for (long long &c : ijk) {
                    char *q = nullptr;
                    c = scan_integer(p, &q);
                    if (q == p || q > end) {
                        out->err = "OBJ: face record is not 'f i j k' (use the tolerant loader)";
                        out->rc = RTX_ERR_UNSUPPORTED;
                        return;
                    }
                    p = q;
                }
                if (last_piece && p >= end) done = true;
            }
            out->faces.insert(out->faces.end(), ijk, ijk + 3);
            out->faces.push_back(static_cast<long long>(out->pos.size() / 3));
        }
    }
}

}  // namespace

namespace {

// Line-based grammar of the tolerant loader for one piece of the text (pieces start at line starts).
void parse_piece_tolerant(const char *p, const char *const end, ObjPiece *out) {
    auto record = [&](const char *q, char id) { return q + 1 < end && q[0] == id && (q[1] == ' ' || q[1] == '\t'); };
    while (p < end) {
        const char *eol = static_cast<const char *>(memchr(p, '\n', static_cast<size_t>(end - p)));
        const char *const next = eol ? eol + 1 : end;
        while (p < next && (*p == ' ' || *p == '\t')) p++;
        if (record(p, 'v')) {
            const char *q = p + 1;
            for (int k = 0; k < 3; k++) {
                char *r = nullptr;
                const double c = scan_double(q, &r);
                if (r == q || r > next) { out->err = "OBJ: vertex record is not 'v x y z'"; out->rc = RTX_ERR_UNSUPPORTED; return; }
                out->pos.push_back(c);
                q = r;
            }
        } else if (record(p, 'f')) {
            const char *q = p + 1;
            for (int k = 0; k < 3; k++) {
                char *r = nullptr;
                const long long c = scan_integer(q, &r);
                if (r == q || r > next) { out->err = "OBJ: face record has fewer than three vertices"; out->rc = RTX_ERR_UNSUPPORTED; return; }
                q = r;
                while (q < next && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') q++;   // the /vt/vn part of the group
                out->faces.push_back(c);
            }
            out->faces.push_back(0);
        }
        p = next;
    }
}

// Tokenises the file in pieces on the worker threads and assembles 9 doubles per face.
// strict: the reference's grammar and its "vertex must be defined before the face" order;
// otherwise the tolerant grammar, indices resolved against the whole vertex list.
int load_obj_text(const std::string &path, bool strict, Doubles *vertices9, std::string *err) {
    StageClock clock;
    FileText text;
    if (!slurp(path, &text)) { *err = "Failed to open file.\n"; return RTX_ERR_IO; }
    clock.lap("read file");
    vertices9->clear();
    const char *const begin = text.data;
    const char *const end = begin + text.size;
    // piece boundaries: a few pieces per thread, at least 256 KiB each
    const size_t want = std::max<size_t>(1, std::min<size_t>(4 * worker_threads(), text.size / (1u << 18) + 1));
    std::vector<const char *> cut{begin};
    for (size_t k = 1; k < want; k++) {
        const char *p = begin + text.size / want * k;
        if (p <= cut.back()) continue;
        while (p + 1 < end && !(p[0] == '\n' && (!strict || p[1] == 'v' || p[1] == 'f'))) p++;
        if (p + 1 >= end) break;
        cut.push_back(p + 1);
    }
    cut.push_back(end);
    const size_t n_pieces = cut.size() - 1;
    std::vector<ObjPiece> pieces(n_pieces);
    parallel_for(n_pieces, [&](size_t k) {
        if (strict) parse_piece(cut[k], cut[k + 1], k + 1 == n_pieces, &pieces[k]);
        else parse_piece_tolerant(cut[k], cut[k + 1], &pieces[k]);
    });
    clock.lap("tokenise pieces");
    for (const ObjPiece &pc : pieces)
        if (pc.rc != RTX_OK) { *err = pc.err; return pc.rc; }
    // global vertex list, face offsets
    std::vector<size_t> v_before(n_pieces + 1, 0), f_before(n_pieces + 1, 0);
    for (size_t k = 0; k < n_pieces; k++) {
        v_before[k + 1] = v_before[k] + pieces[k].pos.size() / 3;
        f_before[k + 1] = f_before[k] + pieces[k].faces.size() / 4;
    }
    const long long n_vertices = static_cast<long long>(v_before[n_pieces]);
    Doubles pos(3 * v_before[n_pieces]);
    parallel_for(n_pieces, [&](size_t k) {
        if (!pieces[k].pos.empty()) memcpy(&pos[3 * v_before[k]], pieces[k].pos.data(), pieces[k].pos.size() * sizeof(double));
    });
    clock.lap("join vertex lists");
    vertices9->resize(9 * f_before[n_pieces]);
    clock.lap("allocate triangles");
    std::vector<int> bad(n_pieces, 0);
    parallel_for(n_pieces, [&](size_t k) {
        const std::vector<long long> &f = pieces[k].faces;
        double *out = vertices9->data() + 9 * f_before[k];
        // strict: the reference indexes vertices[c - 1] unchecked (scene.hpp:237-239); a face may
        // only use vertices that precede it
        const long long defined0 = static_cast<long long>(v_before[k]);
        for (size_t i = 0; i + 3 < f.size(); i += 4, out += 9) {
            const long long limit = strict ? defined0 + f[i + 3] : n_vertices;
            for (int c = 0; c < 3; c++) {
                long long v = f[i + c];
                if (!strict && v < 0) v = n_vertices + v + 1;       // tools/denormalize.py:26-31
                if (v < 1 || v > limit) { bad[k] = 1; return; }
                memcpy(out + 3 * c, &pos[3 * static_cast<size_t>(v - 1)], 3 * sizeof(double));
            }
        }
    });
    clock.lap("resolve faces");
    for (int b : bad)
        if (b) {
            *err = strict ? "OBJ: face references a vertex that is not defined yet" : "OBJ: face references a vertex that does not exist";
            return RTX_ERR_INVALID;
        }
    if (!strict && vertices9->empty()) { *err = "OBJ: no faces"; return RTX_ERR_INVALID; }
    return RTX_OK;
}

}  // namespace

int parse_obj(const std::string &path, Doubles *vertices9, std::string *err) {
    return load_obj_text(path, true, vertices9, err);
}

// tools/denormalize.py:1-36 + Scene::parseOBJ in one pass over the text.  The script
// keeps the lines whose first token is `v` or `f`, writes the vertices first and rewrites
// each face as `f i0 i1 i2` from the leading integer of its first three index groups,
// resolving negative indices against the number of vertices in the whole file.
int parse_obj_tolerant(const std::string &path, Doubles *vertices9, std::string *err) {
    return load_obj_text(path, false, vertices9, err);
}

// Triangle::Triangle, include/triangle.hpp:16-25: the normal (the centroid, which only the host
// BVH builder reads, is computed by fill_centroids when that builder runs).
void setup_triangles(HostScene *hs, bool have_normals) {
    if (have_normals) return;
    const uint64_t n = hs->n_triangles();
    hs->normals.resize(3 * n);
    const uint64_t per = 1u << 15;
    parallel_for((n + per - 1) / per, [&](size_t chunk) {
        for (uint64_t i = chunk * per; i < std::min<uint64_t>(n, (chunk + 1) * per); i++) {
            const double *v = &hs->vertices[9 * i];
            double ax = v[3] - v[0], ay = v[4] - v[1], az = v[5] - v[2];      // e01
            double bx = v[6] - v[0], by = v[7] - v[1], bz = v[8] - v[2];      // e02
            double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
            double len = std::sqrt(cx * cx + cy * cy + cz * cz);
            hs->normals[3 * i] = cx / len;
            hs->normals[3 * i + 1] = cy / len;
            hs->normals[3 * i + 2] = cz / len;
        }
    });
}

namespace {

// centroid = (v0 + v1 + v2) / 3, include/triangle.hpp:24
void fill_centroids(HostScene *hs) {
    const uint64_t n = hs->n_triangles();
    hs->centroids.resize(3 * n);
    const uint64_t per = 1u << 15;
    parallel_for((n + per - 1) / per, [&](size_t chunk) {
        for (uint64_t i = chunk * per; i < std::min<uint64_t>(n, (chunk + 1) * per); i++) {
            const double *v = &hs->vertices[9 * i];
            for (int k = 0; k < 3; k++) hs->centroids[3 * i + k] = ((v[k] + v[3 + k]) + v[6 + k]) / 3;
        }
    });
}

}  // namespace

// The five factories, include/scene.hpp:255-448: they differ only in the file name
// and the camera position.
int apply_preset(HostScene *hs, const std::string &preset, std::string *err) {
    struct P { const char *name; double cam[3]; };
    static const P table[] = {
        {"teapot", {-0.015, 4, -11.99}},           // scene.hpp:259
        {"bunny", {0.1425, 2, -5.94}},             // scene.hpp:298
        {"erato", {-0.8, 28.7, -86.1}},            // scene.hpp:337
        {"dragon", {-0.0425, 0.7115, -3.01725}},   // scene.hpp:376
        {"aurelius", {-0.025, 3.05, -20.9}},       // scene.hpp:415
    };
    // "showcase": not one of the reference's factories.  The teapot's camera and materials, lit by what the shipped
    // scenes never use: two point lights, a directional one, and specular exponents other than 1
    // (scene.hpp:130-135,172-180).
    const bool showcase = preset == "showcase";
    const P *p = nullptr;
    for (const P &e : table) if (preset == e.name || (showcase && std::string(e.name) == "teapot")) p = &e;
    if (!p) { *err = "unknown scene preset '" + preset + "'"; return RTX_ERR_INVALID; }
    const uint64_t n = hs->vertices.size() / 9;
    if (n < 10) {
        // the reference's `size() - 10` wraps around for such a scene
        *err = "scene presets need at least the 10 box triangles";
        return RTX_ERR_INVALID;
    }
    for (int k = 0; k < 3; k++) {
        hs->viewport[k] = 1;
        hs->camera_position[k] = p->cam[k];
        hs->background[k] = 255;
    }
    hs->camera_rotation_deg[0] = 0; hs->camera_rotation_deg[1] = 0.1; hs->camera_rotation_deg[2] = 0;
    hs->ambient = 0.2;
    hs->point_lights.clear();
    hs->directional_lights.assign(1, rtx_light{0.5, {0, 1, -1}});
    // parseOBJ materials (scene.hpp:240-242), then the overrides (scene.hpp:273-289)
    hs->colors.resize(3 * n);
    hs->specularity.resize(n);
    hs->reflectivity.resize(n);
    const uint64_t per = 1u << 16;
    parallel_for((n + per - 1) / per, [&](size_t chunk) {
        for (uint64_t i = chunk * per; i < std::min<uint64_t>(n, (chunk + 1) * per); i++) {
            hs->colors[3 * i] = hs->colors[3 * i + 1] = hs->colors[3 * i + 2] = 255;
            hs->specularity[i] = 1;
            hs->reflectivity[i] = 0.2;
        }
    });
    for (uint64_t i = n - 10; i < n; i++) hs->reflectivity[i] = 0.4;
    for (uint64_t i = n - 4; i < n - 2; i++) { hs->colors[3 * i] = 0; hs->colors[3 * i + 2] = 0; }      // green
    for (uint64_t i = n - 2; i < n; i++) { hs->colors[3 * i + 1] = 0; hs->colors[3 * i + 2] = 0; }      // red
    if (showcase) {
        hs->ambient = 0.15;
        hs->point_lights = {rtx_light{0.6, {2.0, 6.0, -4.0}}, rtx_light{0.3, {-5.0, 3.0, -2.0}}};
        hs->directional_lights.assign(1, rtx_light{0.3, {0, 1, -1}});
        for (uint64_t i = 0; i + 10 < n; i++) {                   // the model: glossy, tinted; the box keeps exponent 1
            hs->specularity[i] = (i % 3 == 0) ? 50 : 10;
            hs->reflectivity[i] = 0.3;
            hs->colors[3 * i] = 230; hs->colors[3 * i + 1] = 200; hs->colors[3 * i + 2] = 120;
        }
    }
    return RTX_OK;
}

// ---------------------------------------------------------------------------------
// BVH::BVH / updateNodeBounds / subdivide / findBestSplitPlane,
// include/bvh.hpp:84-162,236-305, with the recursion replaced by an explicit
// work stack.  Children are allocated pairwise (left = nodesUsed++, right =
// nodesUsed++) when their parent is split and the left subtree is finished before
// the right one is started, so node numbering equals the reference's.
namespace {

struct Builder {
    HostScene &s;
    const double *V;   // vertices
    const double *C;   // centroids
    explicit Builder(HostScene &hs) : s(hs), V(hs.vertices.data()), C(hs.centroids.data()) {}

    // Vector3::min/max (include/vector3.hpp:92-106): keeps `a` unless b is strictly better
    static void grow_point(double lo[3], double hi[3], const double *p) {
        for (int k = 0; k < 3; k++) {
            lo[k] = (lo[k] < p[k]) ? lo[k] : p[k];
            hi[k] = (hi[k] > p[k]) ? hi[k] : p[k];
        }
    }
    // AABB::area (bvh.hpp:74-78)
    static double half_area(const double lo[3], const double hi[3]) {
        double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        return ex * ey + ey * ez + ez * ex;
    }

    void bounds(uint64_t node) {                                        // bvh.hpp:100-115
        rtx_bvh_node &n = s.nodes[node];
        double lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
        // the reference takes min over v0,v1,v2 and then max over v0,v1,v2 per
        // triangle; min and max are independent, so one pass per vertex is the same
        for (uint64_t i = 0; i < n.count; i++) {
            const double *v = V + 9 * s.indices[n.first + i];
            grow_point(lo, hi, v);
            grow_point(lo, hi, v + 3);
            grow_point(lo, hi, v + 6);
        }
        for (int k = 0; k < 3; k++) { n.bmin[k] = lo[k]; n.bmax[k] = hi[k]; }
    }

    double best_plane(const rtx_bvh_node &n, int *axis_out, double *pos_out) const {   // bvh.hpp:236-305
        constexpr int kBins = 8;
        double best = kInf;
        for (int axis = 0; axis < 3; axis++) {
            double cmin = kInf, cmax = -kInf;
            for (uint64_t i = 0; i < n.count; i++) {
                double c = C[3 * s.indices[n.first + i] + axis];
                cmin = (c < cmin) ? c : cmin;                            // std::min(cmin, c)
                cmax = (cmax < c) ? c : cmax;                            // std::max(cmax, c)
            }
            if (cmin == cmax) continue;
            double blo[kBins][3], bhi[kBins][3];
            int bcount[kBins];
            for (int b = 0; b < kBins; b++) {
                bcount[b] = 0;
                for (int k = 0; k < 3; k++) { blo[b][k] = kInf; bhi[b][k] = -kInf; }
            }
            const double scale = kBins / (cmax - cmin);
            for (uint64_t i = 0; i < n.count; i++) {
                const uint64_t t = s.indices[n.first + i];
                size_t b = static_cast<size_t>((C[3 * t + axis] - cmin) * scale);
                if (b > kBins - 1) b = kBins - 1;
                bcount[b]++;
                grow_point(blo[b], bhi[b], V + 9 * t);
                grow_point(blo[b], bhi[b], V + 9 * t + 3);
                grow_point(blo[b], bhi[b], V + 9 * t + 6);
            }
            // prefix boxes from the left and from the right; AABB::grow(AABB) skips a
            // box whose bMin.x is still +inf (bvh.hpp:67-72)
            double larea[kBins - 1], rarea[kBins - 1];
            int lcount[kBins - 1], rcount[kBins - 1];
            double llo[3] = {kInf, kInf, kInf}, lhi[3] = {-kInf, -kInf, -kInf};
            double rlo[3] = {kInf, kInf, kInf}, rhi[3] = {-kInf, -kInf, -kInf};
            int lsum = 0, rsum = 0;
            for (int i = 0; i < kBins - 1; i++) {
                lsum += bcount[i];
                lcount[i] = lsum;
                if (blo[i][0] != kInf) { grow_point(llo, lhi, blo[i]); grow_point(llo, lhi, bhi[i]); }
                larea[i] = half_area(llo, lhi);
                const int rb = kBins - 1 - i;
                rsum += bcount[rb];
                rcount[kBins - 2 - i] = rsum;
                if (blo[rb][0] != kInf) { grow_point(rlo, rhi, blo[rb]); grow_point(rlo, rhi, bhi[rb]); }
                rarea[kBins - 2 - i] = half_area(rlo, rhi);
            }
            const double width = (cmax - cmin) / kBins;
            for (int i = 0; i < kBins - 1; i++) {
                // empty side: 0 * (+inf area) = NaN, and NaN < best is false
                const double cost = lcount[i] * larea[i] + rcount[i] * rarea[i];
                if (cost < best) {
                    best = cost;
                    *axis_out = axis;
                    *pos_out = cmin + width * static_cast<double>(i + 1);
                }
            }
        }
        return best;
    }

    int run(std::string *err) {
        const uint64_t n = s.n_triangles();
        if (n == 0) { *err = "cannot build a BVH over zero triangles"; return RTX_ERR_INVALID; }
        if (n > 0x7fffffffull) { *err = "more than 2^31-1 triangles"; return RTX_ERR_UNSUPPORTED; }
        s.indices.resize(n);
        for (uint64_t i = 0; i < n; i++) s.indices[i] = i;
        s.nodes.assign(2 * n - 1, rtx_bvh_node{{0, 0, 0}, {0, 0, 0}, 0, 0});
        s.nodes[0].first = 0;
        s.nodes[0].count = n;
        uint64_t used = 1;
        bounds(0);
        struct Item { uint64_t node; int32_t depth; };
        std::vector<Item> todo{{0, 0}};
        int32_t deepest = 0;
        while (!todo.empty()) {
            const Item it = todo.back();
            todo.pop_back();
            if (it.depth > deepest) deepest = it.depth;
            rtx_bvh_node &node = s.nodes[it.node];
            int axis = 0;
            double pos = 0;
            const double best = best_plane(node, &axis, &pos);
            // BVHNode::cost (bvh.hpp:80-82)
            if (best >= static_cast<double>(node.count) * half_area(node.bmin, node.bmax)) continue;
            // in-place partition, bvh.hpp:128-140 (ints in the reference)
            int64_t i = static_cast<int64_t>(node.first);
            int64_t j = i + static_cast<int64_t>(node.count) - 1;
            while (i <= j) {
                if (C[3 * s.indices[i] + axis] < pos) {
                    i++;
                } else {
                    const uint64_t keep = s.indices[j];
                    s.indices[j] = s.indices[i];
                    s.indices[i] = keep;
                    j--;
                }
            }
            const uint64_t left_n = static_cast<uint64_t>(i) - node.first;
            if (left_n == 0 || left_n == node.count) continue;
            const uint64_t l = used++, r = used++;
            s.nodes[l].first = node.first;
            s.nodes[l].count = left_n;
            s.nodes[r].first = static_cast<uint64_t>(i);
            s.nodes[r].count = node.count - left_n;
            node.first = l;
            node.count = 0;
            bounds(l);
            bounds(r);
            todo.push_back({r, it.depth + 1});   // popped after the whole left subtree
            todo.push_back({l, it.depth + 1});
        }
        s.nodes.resize(used);
        s.bvh_depth = deepest;
        return RTX_OK;
    }
};

}  // namespace

int build_bvh(HostScene *hs, std::string *err) {
    fill_centroids(hs);         // Triangle ctor work, outside the reference's "[ BVH ]" time as well
    auto t0 = std::chrono::high_resolution_clock::now();
    Builder b(*hs);
    int rc = b.run(err);
    auto t1 = std::chrono::high_resolution_clock::now();
    hs->bvh_seconds = std::chrono::duration<double>(t1 - t0).count();
    return rc;
}

int32_t measure_bvh_depth(const std::vector<rtx_bvh_node> &nodes) {
    if (nodes.empty()) return 0;
    struct Item { uint64_t node; int32_t depth; };
    std::vector<Item> todo{{0, 0}};
    int32_t deepest = 0;
    while (!todo.empty()) {
        Item it = todo.back();
        todo.pop_back();
        if (it.depth > deepest) deepest = it.depth;
        const rtx_bvh_node &n = nodes[it.node];
        if (n.count == 0) {
            todo.push_back({n.first, it.depth + 1});
            todo.push_back({n.first + 1, it.depth + 1});
        }
    }
    return deepest;
}

// A caller-supplied tree is walked once before any kernel or host loop may index with it: child and
// triangle ranges in bounds, every node reached at most once (so the walk terminates), every index a
// valid triangle, the leaves' ranges a partition of the index array (an overlap would make the
// first-leaf-wins-ties rule ambiguous).  Records the walk never reaches -- the reference allocates
// 2n-1 nodes and leaves those past nodesUsed uninitialised (bvh.hpp:91) -- are cut off the end of the
// scene's copy and blanked in between, so that nothing downstream reads them.
int validate_bvh(HostScene *scene, std::string *err) {
    HostScene &hs = *scene;
    const uint64_t n = hs.n_triangles(), m = hs.nodes.size();
    if (m == 0 || hs.indices.size() != n) { *err = "BVH: node or index array has the wrong size"; return RTX_ERR_INVALID; }
    for (uint64_t t : hs.indices) if (t >= n) { *err = "BVH: triangle index out of range"; return RTX_ERR_INVALID; }
    std::vector<uint8_t> seen(m, 0), covered(n, 0);
    std::vector<uint64_t> todo{0};
    uint64_t last = 0, in_leaves = 0;
    while (!todo.empty()) {
        uint64_t k = todo.back();
        todo.pop_back();
        if (seen[k]) { *err = "BVH: a node is reachable twice"; return RTX_ERR_INVALID; }
        seen[k] = 1;
        if (k > last) last = k;
        const rtx_bvh_node &nd = hs.nodes[k];
        if (nd.count == 0) {
            if (nd.first == 0 || nd.first >= m - 1) { *err = "BVH: child index out of range"; return RTX_ERR_INVALID; }
            todo.push_back(nd.first);
            todo.push_back(nd.first + 1);
        } else {
            if (nd.first > n || nd.count > n - nd.first) { *err = "BVH: leaf range out of range"; return RTX_ERR_INVALID; }
            for (uint64_t j = 0; j < nd.count; j++) {
                if (covered[nd.first + j]) { *err = "BVH: leaf ranges overlap"; return RTX_ERR_INVALID; }
                covered[nd.first + j] = 1;
            }
            in_leaves += nd.count;
        }
    }
    if (in_leaves != n) { *err = "BVH: the leaves do not cover every triangle"; return RTX_ERR_INVALID; }
    // Equal-t candidates of different subtrees go to the right one (bvh.hpp:204-208).  The kernels settle such a tie by
    // the leaves' positions in `indices`, which is the same rule only while the leaf ranges ascend in depth-first order
    // (left subtree before right), as the reference's builder lays them out (bvh.hpp:151-154).
    todo.assign(1, 0);
    uint64_t next_first = 0;
    while (!todo.empty()) {
        const rtx_bvh_node &nd = hs.nodes[todo.back()];
        todo.pop_back();
        if (nd.count == 0) {
            todo.push_back(nd.first + 1);
            todo.push_back(nd.first);
        } else {
            if (nd.first != next_first) {
                *err = "BVH: leaf ranges must follow each other in depth-first order (left subtree first), as BVH::subdivide lays them out";
                return RTX_ERR_UNSUPPORTED;
            }
            next_first += nd.count;
        }
    }
    hs.nodes.resize(last + 1);
    for (uint64_t k = 0; k <= last; k++)
        if (!seen[k]) hs.nodes[k] = rtx_bvh_node{{0, 0, 0}, {0, 0, 0}, 0, 0};
    return RTX_OK;
}

}  // namespace rtx
