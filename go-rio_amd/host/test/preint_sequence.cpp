// Drives the drop-in ugpm::VelPreintegration the way Go-RIO's back end does (radar_graph_slam_nodelet.cpp:465-530):
// fill GyroVelData from gyro / ego-velocity queues, PreintOption{type = UGPM}, one inference time, get(0, 0, 0.0, 0.0), read delta_R
// and delta_p.  Input: binary [int n_g][n_g x (t, wx, wy, wz) double][int n_v][n_v x (t, vx, vy, vz) double][start_t][end_t].
// An optional second argument is PreintOption::quantum (chunked mode, preint.h:1584-1702); the chained record is then also rebuilt
// from two half-length constructions with ugpm::combinePreints (math_utils.h:689-726).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <VelInt/preint.h>

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  ugpm::GyroVelData imu_;
  int n = 0;
  if (std::fread(&n, 4, 1, f) != 1) return 2;
  for (int i = 0; i < n; ++i) {
    double r[4];
    if (std::fread(r, 8, 4, f) != 4) return 2;
    ugpm::DataSample s;
    s.t = r[0]; s.data[0] = r[1]; s.data[1] = r[2]; s.data[2] = r[3];
    imu_.gyr_var = 1.74532925e-03;  // RGS:476
    imu_.gyr.push_back(s);
  }
  if (std::fread(&n, 4, 1, f) != 1) return 2;
  for (int i = 0; i < n; ++i) {
    double r[4];
    if (std::fread(r, 8, 4, f) != 4) return 2;
    ugpm::DataSample s;
    s.t = r[0]; s.data[0] = r[1]; s.data[1] = r[2]; s.data[2] = r[3];
    imu_.vel_var = 1e-6;  // RGS:493
    imu_.vel.push_back(s);
  }
  double se[2];
  if (std::fread(se, 8, 2, f) != 2) return 2;
  std::fclose(f);

  ugpm::PreintPrior prior_bias;
  ugpm::PreintOption preint_opt;
  preint_opt.type = ugpm::UGPM;  // RGS:500
  if (argc > 2) preint_opt.quantum = std::atof(argv[2]);
  std::vector<std::vector<double> > t(1, std::vector<double>(1, se[1]));  // RGS:505-508
  try {
    ugpm::VelPreintegration preintegration(imu_, se[0], t, preint_opt, prior_bias, true);  // RGS:512
    ugpm::PreintMeas m = preintegration.get(0, 0, 0.0, 0.0);                                 // RGS:513
    std::printf("{\"delta_R\": [");
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) std::printf("%.17g%s", m.delta_R(r, c), (r == 2 && c == 2) ? "" : ", ");
    std::printf("], \"delta_p\": [%.17g, %.17g, %.17g], \"dt\": %.17g, \"cov00\": %.17g}\n", m.delta_p(0, 0), m.delta_p(1, 0), m.delta_p(2, 0), m.dt, m.cov(0, 0));
    ugpm::PreintMeas mi = preintegration.get(0, 0);  // default bias stds (PRE:55): inflated covariance
    std::printf("{\"cov00_inflated\": %.17g}\n", mi.cov(0, 0));
    try {
      preintegration.get(3, 0);
      return 4;
    } catch (const std::range_error&) {  // PRE:1762
    }
    if (argc > 2) {  // combinePreints through the header: [start, mid] then [mid, end] as two plain constructions
      ugpm::PreintOption plain = preint_opt;
      plain.quantum = -1;
      const double mid = 0.5 * (se[0] + se[1]);
      ugpm::VelPreintegration first(imu_, se[0], mid, plain, prior_bias, true), second(imu_, mid, se[1], plain, prior_bias, true);
      const ugpm::PreintMeas c = ugpm::combinePreints(first.get(0.0, 0.0), second.get(0.0, 0.0));
      std::printf("{\"combined_dt\": %.17g, \"combined_delta_p\": [%.17g, %.17g, %.17g], \"combined_R00\": %.17g}\n", c.dt, c.delta_p(0, 0), c.delta_p(1, 0), c.delta_p(2, 0),
                  c.delta_R(0, 0));
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 3;
  }
  return 0;
}
