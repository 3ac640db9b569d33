// Host-side replay of the sequential association (no GPU work): the native counterpart of
// apse_uav_amd/engines/replay.py::FastReplay for rank 0 of a frame-sharded run, where N x K frames of
// records arrive with one gather and must be turned into track ids + CSV lines without eating the scaling.
// Rules are the reference's (dcnn/engines/rcnn_tracker.py:122-147, dcnn/structures/object_instances.py:48-162,
// dcnn/scripts/tests/visualize_uav.py:117-141); the assignment step is the rectangular shortest-augmenting-
// path solver scipy.optimize.linear_sum_assignment uses (Crouse 2016), in double like scipy.
#include "../../include/apse_hip.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <string>
#include <vector>

namespace {

// Returns col4row for an nr x nc cost matrix with nr <= nc (row-major), minimising the total cost.
std::vector<int> lsap(const std::vector<double>& cost, int nr, int nc) {
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> u(nr, 0.0), v(nc, 0.0), spc(nc);
    std::vector<int> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<char> SR(nr), SC(nc);
    for (int cur = 0; cur < nr; ++cur) {
        double minVal = 0.0;
        int i = cur, sink = -1, nrem = nc;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        std::fill(SR.begin(), SR.end(), 0);
        std::fill(SC.begin(), SC.end(), 0);
        std::fill(spc.begin(), spc.end(), INF);
        while (sink == -1) {
            int index = -1;
            double lowest = INF;
            SR[i] = 1;
            for (int it = 0; it < nrem; ++it) {
                const int j = remaining[it];
                const double r = minVal + cost[(size_t)i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            minVal = lowest;
            if (minVal == INF) return std::vector<int>();
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--nrem];
        }
        u[cur] += minVal;
        for (int r = 0; r < nr; ++r)
            if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= minVal - spc[j];
        int j = sink;
        for (;;) {
            const int r = path[j];
            row4col[j] = r;
            std::swap(col4row[r], j);
            if (r == cur) break;
        }
    }
    return col4row;
}

}  // namespace

struct apse_replay {
    int host_id, edim, max_unseen;
    float thresh;
    int next_id = 1, max_id = 0;
    std::vector<int> ids, since;
    std::vector<float> emb;            // [objects][edim]
};

extern "C" {

apse_replay* apse_replay_create(int host_id, int embed_dim, float dist_thresh, int max_unseen_frames) {
    if (embed_dim < 1 || embed_dim > 4096) return nullptr;
    apse_replay* r = new apse_replay();
    r->host_id = host_id; r->edim = embed_dim; r->thresh = dist_thresh; r->max_unseen = max_unseen_frames;
    return r;
}

void apse_replay_destroy(apse_replay* r) { delete r; }

int apse_replay_max_id(const apse_replay* r) { return r ? r->max_id : -1; }
int apse_replay_next_id(const apse_replay* r) { return r ? r->next_id : -1; }

// One frame.  emb [n][E] f32 (unit vectors), cent [n][2] (1-based, -1 = empty mask), closest [n][n][2].
// Writes the CSV line (NUL-terminated) to line[0..cap) and the track id of every detection to det_ids[n].
// Returns the line length, or a negative APSE_E_* code.
static int replay_step(apse_replay* r, int frame_idx, int n, const float* emb, const int* cent, const int* closest, int cstride,
                       char* line, int cap, int* det_ids) {
    if (!r || n < 0 || !line || cap < 16) return APSE_E_INVALID;
    const int E = r->edim;
    const int O = (int)r->ids.size();
    std::vector<int> det_of(O, -1);                 // object slot -> detection of this frame
    auto add = [&](int d) {
        r->ids.push_back(r->next_id++);
        r->since.push_back(0);
        r->emb.insert(r->emb.end(), emb + (size_t)d * E, emb + (size_t)(d + 1) * E);
        det_of.push_back(d);
    };
    if (n > 0) {
        if (O == 0) {
            for (int d = 0; d < n; ++d) add(d);
        } else {
            // D[o][d] = sum_k (obj[o][k] - det[d][k])^2 in f32 (rcnn_tracker.py:192-221)
            std::vector<float> D((size_t)O * n);
            for (int o = 0; o < O; ++o)
                for (int d = 0; d < n; ++d) {
                    const float* a = &r->emb[(size_t)o * E];
                    const float* b = emb + (size_t)d * E;
                    float s = 0.f;
                    for (int k = 0; k < E; ++k) { const float t = a[k] - b[k]; s += t * t; }
                    D[(size_t)o * n + d] = s;
                }
            // scipy transposes when there are more rows than columns; pairs are the same set either way
            std::vector<char> matched(n, 0);
            std::vector<std::pair<int, int>> pairs;
            if (O <= n) {
                std::vector<double> c((size_t)O * n);
                for (size_t i = 0; i < c.size(); ++i) c[i] = D[i];
                const std::vector<int> col = lsap(c, O, n);
                for (int o = 0; o < (int)col.size(); ++o) pairs.push_back({o, col[o]});
            } else {
                std::vector<double> c((size_t)n * O);
                for (int o = 0; o < O; ++o)
                    for (int d = 0; d < n; ++d) c[(size_t)d * O + o] = D[(size_t)o * n + d];
                const std::vector<int> col = lsap(c, n, O);
                for (int d = 0; d < (int)col.size(); ++d) pairs.push_back({col[d], d});
            }
            for (auto& pr : pairs) {
                const int o = pr.first, d = pr.second;
                if (D[(size_t)o * n + d] < r->thresh) {
                    memcpy(&r->emb[(size_t)o * E], emb + (size_t)d * E, sizeof(float) * E);
                    r->since[o] = 0;
                    det_of[o] = d;
                    matched[d] = 1;
                }
            }
            for (int d = 0; d < n; ++d)
                if (!matched[d]) add(d);
        }
    }
    // drop objects unseen for more than max_unseen frames (counter still holds last frame's value)
    {
        size_t w = 0;
        for (size_t k = 0; k < r->ids.size(); ++k) {
            if (r->since[k] > r->max_unseen) continue;
            if (w != k) {
                r->ids[w] = r->ids[k]; r->since[w] = r->since[k]; det_of[w] = det_of[k];
                memmove(&r->emb[w * E], &r->emb[k * E], sizeof(float) * E);
            }
            ++w;
        }
        r->ids.resize(w); r->since.resize(w); det_of.resize(w); r->emb.resize(w * E);
    }
    int hi = 0, hdet = -1;
    for (size_t k = 0; k < r->ids.size(); ++k) {
        if (det_of[k] >= 0) {
            hi = std::max(hi, r->ids[k]);
            if (r->ids[k] == r->host_id) hdet = det_of[k];
            if (det_ids) det_ids[det_of[k]] = r->ids[k];
            r->since[k] = 0;
        } else {
            r->since[k] += 1;
        }
    }
    r->max_id = std::max(r->max_id, hi);
    if (hi == 0) { line[0] = 0; return 0; }
    std::vector<int> by_id(hi + 1, -1);
    for (size_t k = 0; k < r->ids.size(); ++k)
        if (det_of[k] >= 0) by_id[r->ids[k]] = det_of[k];
    std::string s = std::to_string(frame_idx);
    char buf[64];
    for (int id = 1; id <= hi; ++id) {
        const int d = by_id[id];
        if (d < 0) { s += ",,,,"; continue; }
        const int cx = cent[d * 2], cy = cent[d * 2 + 1];
        if (cx >= 0) { snprintf(buf, sizeof buf, ",%d.0,%d.0", cx, cy); s += buf; } else s += ",nan,nan";
        if (hdet < 0) s += ",nan,nan";
        else {
            const int* c = closest + ((size_t)d * cstride + hdet) * 2;
            snprintf(buf, sizeof buf, ",%d.0,%d.0", c[0], c[1]);
            s += buf;
        }
    }
    if ((int)s.size() + 1 > cap) return APSE_E_INVALID;
    memcpy(line, s.c_str(), s.size() + 1);
    return (int)s.size();
}

int apse_replay_step(apse_replay* r, int frame_idx, int n, const float* emb, const int* cent, const int* closest, char* line,
                     int cap, int* det_ids) {
    return replay_step(r, frame_idx, n, emb, cent, closest, n, line, cap, det_ids);
}

// Replays `nrec` records laid end to end in the wire format of apse_uav_amd/sharding.py::pack_record (count-prefixed f32
// vectors: [0] = n, then boxes [n][4], scores [n], classes [n], centroids [n][2], mass [n], rects [n][4], closest [n][n][2],
// embeddings [n][edim]; `total` floats in all).  Lines are written NUL-free, separated by '\n', into out[0..cap); returns the
// number of bytes written or a negative code.
long long apse_replay_packed(apse_replay* r, const float* recs, long long total, int nrec, int kd, int first_frame, char* out,
                             long long cap) {
    if (!r || (!recs && total > 0) || !out || nrec < 0 || total < 0) return APSE_E_INVALID;
    const int E = r->edim;
    std::vector<int> cent, clos;
    std::vector<char> line(1 << 16);
    long long w = 0, o = 0;
    for (int k = 0; k < nrec; ++k) {
        if (o >= total) return APSE_E_INVALID;
        const float* v = recs + o;
        const int n = (int)v[0];
        if (n < 0 || n > kd) return APSE_E_INVALID;
        const long long len_rec = 1 + (long long)n * 13 + (long long)n * n * 2 + (long long)n * E;
        if (o + len_rec > total) return APSE_E_INVALID;
        const float* vc = v + 1 + n * 6;                 // behind boxes, scores, classes
        const float* vl = v + 1 + n * 13;
        cent.resize((size_t)n * 2 + 1);
        clos.resize((size_t)n * n * 2 + 1);
        for (int i = 0; i < n * 2; ++i) cent[i] = (int)vc[i];
        for (int i = 0; i < n * n * 2; ++i) clos[i] = (int)vl[i];
        const int len = replay_step(r, first_frame + k, n, vl + (size_t)n * n * 2, cent.data(), clos.data(), n, line.data(),
                                    (int)line.size(), nullptr);
        if (len < 0) return len;
        if (w + len + 1 > cap) return APSE_E_INVALID;
        memcpy(out + w, line.data(), len);
        w += len;
        out[w++] = '\n';
        o += len_rec;
    }
    return w;
}

}  // extern "C"
