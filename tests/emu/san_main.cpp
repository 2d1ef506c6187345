#include <cstdio>
#include <vector>
extern "C" int emu_run(const double* chain_model, int n, double* qpos, double* qvel, double* warm, const double* action,
                       int nsub, int debug_env, float* dbgM, float* dbg5, int* counters, double* act);
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  int hdr[5]; fread(hdr, sizeof(int), 5, f);   // ncm, nv, nu, na, ncases
  std::vector<double> cm(hdr[0]); fread(cm.data(), sizeof(double), hdr[0], f);
  int nv = hdr[1], nu = hdr[2], na = hdr[3];
  for (int i = 0; i < hdr[4]; i++) {
    std::vector<double> q(nv), v(nv), w(nv, 0.0), a(nu), act(na > 0 ? na : 1, 0.0);
    fread(q.data(), sizeof(double), nv, f); fread(v.data(), sizeof(double), nv, f); fread(a.data(), sizeof(double), nu, f);
    int cnt[16] = {0};   // emu_run reports 8 counters
    int rc = emu_run(cm.data(), 1, q.data(), v.data(), w.data(), a.data(), 3, -1, nullptr, nullptr, cnt, na > 0 ? act.data() : nullptr);
    printf("case %d rc %d iters %d ncon %d q0 %.6f\n", i, rc, cnt[0], cnt[3], q[2]);
  }
  return 0;
}
