// uncompress -- command-line front end with the reference's flag surface
// (uncompress.cpp:53-65): wavelet ('B') and Huffman ('H') streams, inverse BWT on the MI355X.
//   uncompress [-i] [-c] [-d device] [-v N] [input] [output]
#include <getopt.h>
#include <chrono>
#include <string>

#include "bwtc_hip_decode.hpp"

int main(int argc, char** argv) {
  bool from_stdin = false, to_stdout = false;
  int verbosity = 0, device = 0;
  static option longopts[] = {{"stdin", no_argument, 0, 'i'}, {"stdout", no_argument, 0, 'c'},
                              {"device", required_argument, 0, 'd'}, {"verb", required_argument, 0, 'v'},
                              {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
  int o;
  while ((o = getopt_long(argc, argv, "icd:v:h", longopts, 0)) != -1) {
    switch (o) {
      case 'i': from_stdin = true; break;
      case 'c': to_stdout = true; break;
      case 'd': device = std::atoi(optarg); break;
      case 'v': verbosity = std::atoi(optarg); break;
      default:
        std::fprintf(stderr, "usage: uncompress [-i] [-c] [input] [output]\n");
        return o == 'h' ? 0 : 1;
    }
  }
  std::string in_name, out_name;
  if (!from_stdin && optind < argc) in_name = argv[optind++];
  if (!to_stdout) {
    if (optind < argc) out_name = argv[optind++];
    else if (in_name.size() > 5 && in_name.substr(in_name.size() - 5) == ".bwtc") out_name = in_name.substr(0, in_name.size() - 5);
    else if (!in_name.empty()) out_name = in_name + ".out";
  }
  if (!from_stdin && in_name.empty()) { std::fprintf(stderr, "no input\n"); return 1; }
  const auto t0 = std::chrono::steady_clock::now();
  bwtc::Decompressor d(new bwtc::RawInStream(in_name), new bwtc::RawOutStream(out_name), device);
  const size_t n = d.decompress(1);
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (verbosity > 0) std::fprintf(stderr, "Decompressed size: %zu bytes, %.3f s\n", n, s);
  return 0;
}
