/* ORACLE (test infrastructure only -- see oracle/__init__.py).
 *
 * Plain-C restatement of the reference's native plug-in `pafprocess` (COCO-18 topology):
 *   process_paf            third_party_methods/lib/pafprocess/pafprocess.cpp:22-194
 *   getters                third_party_methods/lib/pafprocess/pafprocess.cpp:196-218
 *   get_paf_vectors        third_party_methods/lib/pafprocess/pafprocess.cpp:220-239
 *   roundpaf/comp          third_party_methods/lib/pafprocess/pafprocess.cpp:241-247
 *   constants, topology    third_party_methods/lib/pafprocess/pafprocess.h:6-24
 *
 * Pinned: oracle/Makefile compiles the reference's own pafprocess.cpp (as it lies under
 * /root/reference) into oracle/_ref/libpafprocess_ref.so and tests/test_oracle_pafprocess.py
 * checks this restatement against it on seeded inputs, and both against the committed golden
 * vectors.  One deliberate difference: std::sort is unstable, this restatement (like the HIP
 * kernel) orders equal-score candidates by candidate index; seeded float scores never tie.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off (no fused multiply-add, like the x86-64 g++
 * build of the reference).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NUM_PART 18
#define NUM_PAIR 19
#define STEP_PAF 10
#define MAX_PEAK 4096
#define MAX_HUMAN 1024

static const float THRESH_VECTOR_SCORE = 0.05;
static const int THRESH_VECTOR_CNT1 = 6;
static const int THRESH_PART_CNT = 4;
static const float THRESH_HUMAN_SCORE = 0.3;

static const int PAIRS_NET[NUM_PAIR][2] = {
    {12, 13}, {20, 21}, {14, 15}, {16, 17}, {22, 23}, {24, 25}, {0, 1}, {2, 3}, {4, 5}, {6, 7},
    {8, 9}, {10, 11}, {28, 29}, {30, 31}, {34, 35}, {32, 33}, {36, 37}, {18, 19}, {26, 27}};
static const int PAIRS[NUM_PAIR][2] = {
    {1, 2}, {1, 5}, {2, 3}, {3, 4}, {5, 6}, {6, 7}, {1, 8}, {8, 9}, {9, 10}, {1, 11},
    {11, 12}, {12, 13}, {1, 0}, {0, 14}, {14, 16}, {0, 15}, {15, 17}, {2, 16}, {5, 17}};

typedef struct { int x, y; float score; int id; } peak_t;
typedef struct { int ia, ib; float score; int order; } cand_t;
typedef struct { int cid1, cid2; float score; int ia, ib; } conn_t;

static peak_t g_line[MAX_PEAK];
static int g_nline;
static float g_rows[MAX_HUMAN][20];
static int g_nrows;

static int cmp_cand(const void *a, const void *b) {
    const cand_t *x = (const cand_t *)a, *y = (const cand_t *)b;
    if (x->score > y->score) return -1;
    if (x->score < y->score) return 1;
    return x->order - y->order;
}

int oracle_process_paf(int p1, int p2, int p3, const float *peaks, int h1, int h2, int h3, const float *heatmap,
                       int f1, int f2, int f3, const float *pafmap) {
    (void)h2; (void)h3; (void)heatmap; (void)f1;
    static peak_t part[NUM_PART][MAX_PEAK / 4];
    int npart[NUM_PART];
    memset(npart, 0, sizeof npart);
    int cnt = 0;
    for (int i = 0; i < p1; ++i)
        for (int k = 0; k < p2; ++k) {
            const float *r = peaks + ((size_t)i * p2 + k) * p3;
            peak_t pk;
            pk.id = cnt++;
            pk.x = (int)r[0];
            pk.y = (int)r[1];
            pk.score = r[2];
            int pid = (int)r[4];
            if (pid < 0 || pid >= NUM_PART || npart[pid] >= MAX_PEAK / 4) return -1;
            part[pid][npart[pid]++] = pk;
        }
    g_nline = 0;
    for (int p = 0; p < NUM_PART; ++p)
        for (int i = 0; i < npart[p]; ++i) g_line[g_nline++] = part[p][i];

    static conn_t conns[NUM_PAIR][MAX_PEAK / 4];
    int nconn[NUM_PAIR];
    for (int pair = 0; pair < NUM_PAIR; ++pair) {
        nconn[pair] = 0;
        const int pa = PAIRS[pair][0], pb = PAIRS[pair][1];
        const int na = npart[pa], nb = npart[pb];
        if (na == 0 || nb == 0) continue;
        cand_t *cands = (cand_t *)malloc(sizeof(cand_t) * (size_t)na * nb);
        int nc = 0;
        for (int ia = 0; ia < na; ++ia)
            for (int ib = 0; ib < nb; ++ib) {
                const peak_t a = part[pa][ia], b = part[pb][ib];
                float vx = (float)(b.x - a.x), vy = (float)(b.y - a.y);
                float norm = (float)sqrt(vx * vx + vy * vy);
                if (norm < 1e-12) continue;
                vx = vx / norm;
                vy = vy / norm;
                const float stepx = (b.x - a.x) / (float)STEP_PAF, stepy = (b.y - a.y) / (float)STEP_PAF;
                float scores = 0.0f;
                int c1 = 0;
                for (int i = 0; i < STEP_PAF; ++i) {
                    int lx = (int)(a.x + i * stepx + 0.5);
                    int ly = (int)(a.y + i * stepy + 0.5);
                    float px = pafmap[PAIRS_NET[pair][0] + f3 * (lx + f2 * ly)];
                    float py = pafmap[PAIRS_NET[pair][1] + f3 * (lx + f2 * ly)];
                    float s = vx * px + vy * py;
                    scores += s;
                    if (s > THRESH_VECTOR_SCORE) c1 += 1;
                }
                float c2 = scores / STEP_PAF + fmin(0.0, 0.5 * h1 / norm - 1.0);
                if (c1 > THRESH_VECTOR_CNT1 && c2 > 0) {
                    cands[nc].ia = ia; cands[nc].ib = ib; cands[nc].score = c2; cands[nc].order = nc;
                    ++nc;
                }
            }
        qsort(cands, nc, sizeof(cand_t), cmp_cand);
        for (int c = 0; c < nc; ++c) {
            int taken = 0;
            for (int k = 0; k < nconn[pair]; ++k)
                if (conns[pair][k].ia == cands[c].ia || conns[pair][k].ib == cands[c].ib) { taken = 1; break; }
            if (taken) continue;
            conn_t cn;
            cn.ia = cands[c].ia; cn.ib = cands[c].ib; cn.score = cands[c].score;
            cn.cid1 = part[pa][cn.ia].id; cn.cid2 = part[pb][cn.ib].id;
            conns[pair][nconn[pair]++] = cn;
        }
        free(cands);
    }

    g_nrows = 0;
    for (int pair = 0; pair < NUM_PAIR; ++pair) {
        const int q1 = PAIRS[pair][0], q2 = PAIRS[pair][1];
        for (int c = 0; c < nconn[pair]; ++c) {
            const conn_t cn = conns[pair][c];
            int found = 0, i1 = 0, i2 = 0;
            for (int s = 0; s < g_nrows; ++s)
                if (g_rows[s][q1] == cn.cid1 || g_rows[s][q2] == cn.cid2) {
                    if (found == 0) i1 = s;
                    if (found == 1) i2 = s;
                    found += 1;
                }
            if (found == 1) {
                if (g_rows[i1][q2] != cn.cid2) {
                    g_rows[i1][q2] = cn.cid2;
                    g_rows[i1][19] += 1;
                    g_rows[i1][18] += g_line[cn.cid2].score + cn.score;
                }
            } else if (found == 2) {
                int membership = 0;
                for (int s = 0; s < 18; ++s)
                    if (g_rows[i1][s] > 0 && g_rows[i2][s] > 0) membership = 2;
                if (membership == 0) {
                    for (int s = 0; s < 18; ++s) g_rows[i1][s] += (g_rows[i2][s] + 1);
                    g_rows[i1][19] += g_rows[i2][19];
                    g_rows[i1][18] += g_rows[i2][18];
                    g_rows[i1][18] += cn.score;
                    memmove(g_rows[i2], g_rows[i2 + 1], sizeof(float) * 20 * (size_t)(g_nrows - 1 - i2));
                    --g_nrows;
                } else {
                    g_rows[i1][q2] = cn.cid2;
                    g_rows[i1][19] += 1;
                    g_rows[i1][18] += g_line[cn.cid2].score + cn.score;
                }
            } else if (found == 0 && pair < 18) {
                if (g_nrows >= MAX_HUMAN) return -2;
                float *row = g_rows[g_nrows++];
                for (int s = 0; s < 20; ++s) row[s] = -1;
                row[q1] = cn.cid1;
                row[q2] = cn.cid2;
                row[19] = 2;
                row[18] = g_line[cn.cid1].score + g_line[cn.cid2].score + cn.score;
            }
        }
    }
    for (int i = g_nrows - 1; i >= 0; --i)
        if (g_rows[i][19] < THRESH_PART_CNT || g_rows[i][18] / g_rows[i][19] < THRESH_HUMAN_SCORE) {
            memmove(g_rows[i], g_rows[i + 1], sizeof(float) * 20 * (size_t)(g_nrows - 1 - i));
            --g_nrows;
        }
    return 0;
}

int oracle_get_num_humans(void) { return g_nrows; }
int oracle_get_part_cid(int human_id, int part_id) { return (int)g_rows[human_id][part_id]; }
float oracle_get_score(int human_id) { return g_rows[human_id][18] / g_rows[human_id][19]; }
int oracle_get_part_x(int cid) { return g_line[cid].x; }
int oracle_get_part_y(int cid) { return g_line[cid].y; }
float oracle_get_part_score(int cid) { return g_line[cid].score; }
