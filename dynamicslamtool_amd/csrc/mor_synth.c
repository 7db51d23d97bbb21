/*
 * mor_synth.c — deterministic synthetic LiDAR streams.  Measurement and test
 * infrastructure: there is no KITTI data and no network in the build or GPU containers.
 *
 *   sensor 0  hdl64 : 64 beams, elevation +2.0° … −24.8°, 1875 azimuth steps → 120 000 pts
 *   sensor 1  os128 : 128 beams, ±22.5°, 2048 azimuth steps               → 262 144 pts
 *   sensor 2  agg10 : 10 consecutive hdl64 sweeps merged in the last pose, truncated to 1 000 000
 *   sensor 3  hdl64_urban : the hdl64 sensor in a street scene of realistic density (below)
 *
 * Scene (per seed): ground plane 1.73 m below the sensor, 20–40 yawed boxes (cars 4×1.8×1.5,
 * pedestrians 0.6×0.6×1.7, walls 20×0.3×3) uniformly in ±40 m, 3–6 of the cars/pedestrians moving
 * 0.3–1.0 m per frame along their heading.  Ego pose at frame f: f·1.0 m along the heading, which
 * turns 1° per frame.  Range noise σ = 2 cm.  Rays without a hit within 120 m become far ground
 * returns, so every frame has exactly the nominal point count.  Points are in the sensor frame,
 * fp32 (x,y,z,intensity); pose = sensor position + yaw quaternion in the world frame.
 *
 * Urban scene (sensor 3): the open scene above puts ~90 % of a sweep on the ground plane.  A real urban KITTI sweep is the
 * opposite — most returns come from façades, vegetation, parked cars — so this scene has: a street along the ego
 * direction lined by building façades 7–12 m to either side (boxes 15–40 m long with gaps and set-backs, cross streets every
 * ~60 m, a second row of buildings and garden walls behind the gaps), kerbs (0.35 m high strips), trees (trunk + 50–90 leaf cubes of 0.25–0.5 m scattered in the crown: the irregular
 * returns of vegetation), hedges of scattered cubes, poles, 12–20 parked cars, pedestrians, and 3–6 moving cars /
 * pedestrians.  Ego motion: 1 m per frame, heading turning 0.15° per frame.  Roughly half of the trimmed sweep is non-ground.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static inline double urand(uint64_t *s) { return (double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0); }
static inline double nrand(uint64_t *s) { double u1 = urand(s), u2 = urand(s); if (u1 < 1e-300) u1 = 1e-300; return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2); }

typedef struct { double cx, cy, cz, hx, hy, hz, yaw, vx, vy; } sbox;
#define MAXB 4096
#define MAXC 1024
typedef struct { int n; double yaw_rate_deg; sbox b[MAXB]; } scene;

static void make_scene(uint64_t seed, scene *sc) {
  uint64_t s = seed * 0x2545F4914F6CDD1Dull + 12345;
  sc->yaw_rate_deg = 1.0;
  sc->n = 20 + (int)(splitmix(&s) % 21);
  int n_moving_target = 3 + (int)(splitmix(&s) % 4), n_moving = 0;
  for (int i = 0; i < sc->n; ++i) {
    sbox *b = &sc->b[i]; double t = urand(&s);
    double sx, sy, sz; int movable = 1;
    if (t < 0.5) { sx = 4.0; sy = 1.8; sz = 1.5; } else if (t < 0.8) { sx = 0.6; sy = 0.6; sz = 1.7; } else { sx = 20.0; sy = 0.3; sz = 3.0; movable = 0; }
    do { b->cx = (urand(&s) * 2 - 1) * 40.0; b->cy = (urand(&s) * 2 - 1) * 40.0; } while (fabs(b->cy) < 3.0 && b->cx > -5.0); /* keep the ego lane clear */
    b->hx = sx / 2; b->hy = sy / 2; b->hz = sz / 2; b->cz = sz / 2; /* world z = 0 is the ground */
    b->yaw = urand(&s) * 6.283185307179586; b->vx = b->vy = 0;
    if (movable && n_moving < n_moving_target) { double v = 0.3 + 0.7 * urand(&s); b->vx = v * cos(b->yaw); b->vy = v * sin(b->yaw); ++n_moving; }
  }
}

static void add_box(scene *sc, double cx, double cy, double cz, double sx, double sy, double sz, double yaw, double v) {
  if (sc->n >= MAXB) return;
  sbox *b = &sc->b[sc->n++];
  b->cx = cx; b->cy = cy; b->cz = cz; b->hx = sx / 2; b->hy = sy / 2; b->hz = sz / 2; b->yaw = yaw; b->vx = v * cos(yaw); b->vy = v * sin(yaw);
}
static void make_scene_urban(uint64_t seed, scene *sc) {
  uint64_t s = seed * 0x2545F4914F6CDD1Dull + 99991;
  sc->n = 0; sc->yaw_rate_deg = 0.15;
  const double PI = 3.14159265358979323846;
  for (int side = -1; side <= 1; side += 2) {
    /* façades: a broken line of buildings along x from −70 to +110 m */
    double x = -70.0 + 10.0 * urand(&s);
    while (x < 110.0) {
      double len = 15.0 + 25.0 * urand(&s), setback = 7.0 + 5.0 * urand(&s), depth = 8.0 + 6.0 * urand(&s), h = 6.0 + 9.0 * urand(&s);
      if (fmod(x + 70.0, 60.0) > 48.0) { x += 12.0; continue; }                 /* cross street */
      add_box(sc, x + len / 2, side * (setback + depth / 2), h / 2, len, depth, h, (urand(&s) - 0.5) * 0.06, 0);
      if (urand(&s) < 0.5) add_box(sc, x + len * urand(&s), side * (setback - 0.4), 1.2, 1.5 + 2.0 * urand(&s), 0.8, 2.4, 0, 0);   /* porch / bay */
      x += len + (urand(&s) < 0.6 ? 4.0 + 10.0 * urand(&s) : 0.0);
    }
    /* a second row of buildings and garden walls seen through the gaps */
    for (double bx = -70.0 + 15.0 * urand(&s); bx < 110.0; bx += 18.0 + 14.0 * urand(&s)) {
      double len = 8.0 + 12.0 * urand(&s), dist = 22.0 + 14.0 * urand(&s);
      add_box(sc, bx, side * dist, 4.0, len, 7.0, 8.0, (urand(&s) - 0.5) * 0.5, 0);
      add_box(sc, bx + 6.0 * (urand(&s) - 0.5), side * (dist - 6.0 - 3.0 * urand(&s)), 0.9, 5.0 + 6.0 * urand(&s), 0.25, 1.8, (urand(&s) - 0.5) * 1.0, 0);
    }
    /* kerb strips, 0.35 m high, 5 m from the lane centre */
    for (double kx = -70.0; kx < 110.0; kx += 30.0) add_box(sc, kx + 15.0, side * 5.0, 0.175, 30.0, 0.3, 0.35, 0, 0);
    /* trees every 9–15 m between kerb and façade */
    for (double tx = -60.0 + 8.0 * urand(&s); tx < 100.0; tx += 6.0 + 5.0 * urand(&s)) {
      double ty = side * (urand(&s) < 0.6 ? 5.8 + 0.8 * urand(&s) : 13.0 + 14.0 * urand(&s)), th = 1.6 + 1.2 * urand(&s), cr = 1.4 + 1.2 * urand(&s);
      add_box(sc, tx, ty, th / 2, 0.3, 0.3, th, urand(&s) * PI, 0);
      int leaves = 70 + (int)(splitmix(&s) % 61);
      for (int l = 0; l < leaves; ++l) {
        double u = urand(&s) * 2 - 1, ph = urand(&s) * 2 * PI, rr = cr * cbrt(urand(&s)), q = sqrt(1 - u * u), e = 0.25 + 0.25 * urand(&s);
        add_box(sc, tx + rr * q * cos(ph), ty + rr * q * sin(ph), th + 0.3 * cr + 0.8 * rr * u, e, e, e, urand(&s) * PI, 0);
      }
    }
    /* hedges: rows of scattered cubes 0.5–1.3 m high in front of some façades */
    for (int hdg = 0; hdg < 9; ++hdg) {
      double hx = -50.0 + 140.0 * urand(&s), hl = 6.0 + 10.0 * urand(&s), hy = side * (hdg < 4 ? 6.3 + 0.4 * urand(&s) : 12.0 + 20.0 * urand(&s));
      int n = (int)(hl * 6);
      for (int l = 0; l < n; ++l) { double e = 0.3 + 0.3 * urand(&s); add_box(sc, hx + hl * urand(&s), hy + 0.5 * (urand(&s) - 0.5), 0.3 + 0.9 * urand(&s), e, e, e, urand(&s) * PI, 0); }
    }
    /* poles and parked cars */
    for (double px = -60.0 + 20.0 * urand(&s); px < 100.0; px += 25.0 + 10.0 * urand(&s)) add_box(sc, px, side * 5.4, 3.0, 0.2, 0.2, 6.0, 0, 0);
    int cars = 6 + (int)(splitmix(&s) % 5);
    for (int c = 0; c < cars; ++c) add_box(sc, -50.0 + 140.0 * urand(&s), side * (3.4 + 0.3 * urand(&s)), 0.75, 4.0 + 0.6 * urand(&s), 1.8, 1.5, (urand(&s) - 0.5) * 0.1, 0);
    int peds = 3 + (int)(splitmix(&s) % 4);
    for (int c = 0; c < peds; ++c) add_box(sc, -30.0 + 100.0 * urand(&s), side * (5.6 + 1.0 * urand(&s)), 0.85, 0.6, 0.6, 1.7, urand(&s) * PI, 0);
  }
  /* movers: cars on the lanes, pedestrians along the pavements */
  int movers = 3 + (int)(splitmix(&s) % 4);
  for (int m = 0; m < movers; ++m) {
    if (urand(&s) < 0.6) add_box(sc, 8.0 + 50.0 * urand(&s), (urand(&s) < 0.5 ? -1.7 : 1.7), 0.75, 4.0, 1.8, 1.5, (urand(&s) < 0.5 ? 0.0 : PI), 0.4 + 0.6 * urand(&s));
    else add_box(sc, -10.0 + 50.0 * urand(&s), (urand(&s) < 0.5 ? -6.2 : 6.2), 0.85, 0.6, 0.6, 1.7, (urand(&s) < 0.5 ? 0.0 : PI), 0.3 + 0.3 * urand(&s));
  }
}

/* one sweep at frame index f into out (n_beams*n_az points); returns pose */
static void sweep(const scene *sc, uint64_t seed, int f, int n_beams, int n_az, double el_top_deg, double el_bot_deg, float *out, double pose[7]) {
  const double H = 1.73, DEG = 3.14159265358979323846 / 180.0;
  /* ego trajectory: heading turns 1° per frame, 1 m per frame */
  double ex = 0, ey = 0, eyaw = 0;
  for (int k = 0; k < f; ++k) { ex += cos(eyaw); ey += sin(eyaw); eyaw += sc->yaw_rate_deg * DEG; }
  pose[0] = ex; pose[1] = ey; pose[2] = H; pose[3] = 0; pose[4] = 0; pose[5] = sin(eyaw / 2); pose[6] = cos(eyaw / 2);
  uint64_t s = seed * 0xD1342543DE82EF95ull + (uint64_t)f * 0x9E3779B97F4A7C15ull + 777;
  /* box state at frame f, in the sensor frame's yaw-aligned coordinates we keep world and rotate rays */
  static __thread double bcx[MAXB], bcy[MAXB], bc[MAXB], bs[MAXB], rad[MAXB];
  for (int i = 0; i < sc->n; ++i) { const sbox *b = &sc->b[i]; bcx[i] = b->cx + b->vx * f - ex; bcy[i] = b->cy + b->vy * f - ey; bc[i] = cos(b->yaw); bs[i] = sin(b->yaw); rad[i] = sqrt(b->hx * b->hx + b->hy * b->hy); }
  size_t o = 0;
  for (int a = 0; a < n_az; ++a) {
    double az = eyaw + (double)a * (6.283185307179586 / n_az), ca = cos(az), sa = sin(az);
    /* cull boxes by perpendicular distance of their centre to this azimuth's vertical plane */
    int cand[MAXC], nc = 0;
    for (int i = 0; i < sc->n && nc < MAXC; ++i) { double along = bcx[i] * ca + bcy[i] * sa, perp = fabs(-bcx[i] * sa + bcy[i] * ca); if (perp <= rad[i] && along > -rad[i]) cand[nc++] = i; }
    double la = (double)a * (6.283185307179586 / n_az), lca = cos(la), lsa = sin(la); /* azimuth in the sensor frame */
    for (int bm = 0; bm < n_beams; ++bm) {
      double el = (el_top_deg + (el_bot_deg - el_top_deg) * (double)bm / (double)(n_beams - 1)) * DEG, ce = cos(el), se = sin(el);
      double dx = ce * ca, dy = ce * sa, dz = se; /* world direction; origin (0,0,H) relative to ego xy */
      double best = 1e30;
      if (dz < -1e-9) { double t = -H / dz; if (t < best) best = t; }
      for (int c = 0; c < nc; ++c) { int i = cand[c]; const sbox *b = &sc->b[i];
        /* ray in box frame */
        double ox = -bcx[i], oy = -bcy[i], oz = H - b->cz;
        double rox = ox * bc[i] + oy * bs[i], roy = -ox * bs[i] + oy * bc[i];
        double rdx = dx * bc[i] + dy * bs[i], rdy = -dx * bs[i] + dy * bc[i];
        double t0 = 0, t1 = best; int ok = 1; double O[3] = { rox, roy, oz }, D[3] = { rdx, rdy, dz }, Hh[3] = { b->hx, b->hy, b->hz };
        for (int d = 0; d < 3 && ok; ++d) {
          if (fabs(D[d]) < 1e-12) { if (fabs(O[d]) > Hh[d]) ok = 0; }
          else { double ta = (-Hh[d] - O[d]) / D[d], tb = (Hh[d] - O[d]) / D[d]; if (ta > tb) { double tmp = ta; ta = tb; tb = tmp; } if (ta > t0) t0 = ta; if (tb < t1) t1 = tb; if (t0 > t1) ok = 0; } }
        if (ok && t0 > 0.5 && t0 < best) best = t0; }
      double r;
      if (best > 120.0) { /* no return: far ground hit along this azimuth */
        double g = 50.0 + 70.0 * urand(&s); out[4 * o] = (float)(g * lca); out[4 * o + 1] = (float)(g * lsa); out[4 * o + 2] = (float)(-H + 0.02 * nrand(&s)); out[4 * o + 3] = (float)urand(&s); ++o; continue; }
      r = best + 0.02 * nrand(&s);
      out[4 * o] = (float)(r * ce * lca); out[4 * o + 1] = (float)(r * ce * lsa); out[4 * o + 2] = (float)(r * se); out[4 * o + 3] = (float)urand(&s); ++o;
    }
  }
}

static void quat_yaw(const double p[7], double *yaw) { *yaw = 2.0 * atan2(p[5], p[6]); }

static int mor_synth_frame_scene(const scene *scp, uint64_t seed, int sensor, int frame_idx, float *out_xyzi, double pose7[7]);
uint64_t mor_synth_points(int sensor) { return (sensor == 0 || sensor == 3) ? 120000u : sensor == 1 ? 262144u : 1000000u; }

/* Fill out_xyzi (mor_synth_points(sensor)*4 floats) and pose7 for frame `frame_idx` of stream `seed`. */
int mor_synth_frame(uint64_t seed, int sensor, int frame_idx, float *out_xyzi, double pose7[7]) {
  scene *scp = (scene *)malloc(sizeof(scene)); if (!scp) return -1;
  if (sensor == 3) make_scene_urban(seed, scp); else make_scene(seed, scp);
  int rc_ = mor_synth_frame_scene(scp, seed, sensor, frame_idx, out_xyzi, pose7); free(scp); return rc_;
}
static int mor_synth_frame_scene(const scene *scp, uint64_t seed, int sensor, int frame_idx, float *out_xyzi, double pose7[7]) {
#define sc (*scp)
  if (sensor == 3) { sweep(&sc, seed, frame_idx, 64, 1875, 2.0, -24.8, out_xyzi, pose7); return 0; }
  if (sensor == 0) { sweep(&sc, seed, frame_idx, 64, 1875, 2.0, -24.8, out_xyzi, pose7); return 0; }
  if (sensor == 1) { sweep(&sc, seed, frame_idx, 128, 2048, 22.5, -22.5, out_xyzi, pose7); return 0; }
  if (sensor == 2) {
    const size_t per = 120000, total = 1000000; float *tmp = (float *)malloc(per * 4 * sizeof(float)); double last[7], p[7];
    sweep(&sc, seed, frame_idx + 9, 64, 1875, 2.0, -24.8, tmp, last); double lyaw; quat_yaw(last, &lyaw);
    size_t o = 0;
    for (int k = 0; k < 10 && o < total; ++k) {
      sweep(&sc, seed, frame_idx + k, 64, 1875, 2.0, -24.8, tmp, p); double yaw; quat_yaw(p, &yaw);
      double dyaw = yaw - lyaw, c = cos(dyaw), s = sin(dyaw), tx = p[0] - last[0], ty = p[1] - last[1];
      double cl = cos(-lyaw), sl = sin(-lyaw), rx = tx * cl - ty * sl, ry = tx * sl + ty * cl; /* translation in the last sensor frame */
      for (size_t i = 0; i < per && o < total; ++i, ++o) { double x = tmp[4 * i], y = tmp[4 * i + 1];
        out_xyzi[4 * o] = (float)(x * c - y * s + rx); out_xyzi[4 * o + 1] = (float)(x * s + y * c + ry); out_xyzi[4 * o + 2] = tmp[4 * i + 2]; out_xyzi[4 * o + 3] = tmp[4 * i + 3]; }
    }
    memcpy(pose7, last, 7 * sizeof(double)); free(tmp); return 0;
  }
  return -1;
#undef sc
}

/* n_frames frames laid out back to back; seeds[i], frame_idx[i] per frame; OpenMP over frames. */
int mor_synth_batch(int sensor, int n_frames, const uint64_t *seeds, const int *frame_idx, float *out_xyzi, double *pose7) {
  uint64_t np = mor_synth_points(sensor); int rc = 0;
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < n_frames; ++i) { if (mor_synth_frame(seeds[i], sensor, frame_idx[i], out_xyzi + (size_t)i * np * 4, pose7 + 7 * (size_t)i)) rc = -1; }
  return rc;
}

#ifndef MOR_SRC_HASH_STR
#define MOR_SRC_HASH_STR "MOR_SRC_HASH=unknown"
#endif
const char *mor_synth_build_hash(void) { return MOR_SRC_HASH_STR; }
