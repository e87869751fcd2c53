// compress -- command-line front end with the reference's flag surface (compress.cpp:107-145)
// on top of the MI355X back-end.  Output files decode with the reference's `uncompress`.
//
//   compress [options] [input] [output]
//     -m, --mem MB       memory budget; BWT block = 0.185 * MB * 1e6 bytes   (default 100)
//     -s, --starts N     starting points for the inverse transform, 1..256    (default 8)
//         --bwt C        BWT algorithm: g (GPU) or a (auto = g)              (default a)
//     -e, --enc C        entropy coder: B (wavelet, the reference's default), b, u (wavelet with
//                        other main models) or H (Huffman)                    (default B)
//     -i, --stdin        read from standard input
//     -c, --stdout       write to standard output
//     -d, --device N     GPU to use                                          (default 0)
//     -D, --devices LIST farm the blocks of the stream over these GPUs, e.g. 0,1,2,3 (one context,
//                        one thread and one page-locked staging ring per entry; the same bytes as
//                        with one device)
//     -P, --pipeline N   blocks under way per context for the wavelet coders (default: 128 or 96
//                        when the blocks are 64 MB and more, the input file has 256 blocks and
//                        more per context and the host has 40 / 24 GB free, else 16;
//                        BWTC_HIP_WAVELET_DEPTH overrides).  From 56 on the library codes
//                        with its fused 16-lane engines: a block then takes seconds, the stream
//                        two to three times less host time.
//     -v, --verb N       verbosity
// Differences from the reference, on purpose: --bwt d/s (CPU back-ends) and --enc m/M (models
// whose reference implementation reads past its table) are rejected, 'm' / 'M' are not offered.
#include <getopt.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "bwtc_hip.hpp"
#include "bwtc_hip_farm.hpp"

static double memAvailableGB() {
  double gb = 0.0;
  if (FILE* f = std::fopen("/proc/meminfo", "r")) {
    char line[256];
    while (std::fgets(line, sizeof line, f)) {
      unsigned long long kb;
      if (std::sscanf(line, "MemAvailable: %llu kB", &kb) == 1) { gb = kb / 1048576.0; break; }
    }
    std::fclose(f);
  }
  return gb;
}

int main(int argc, char** argv) {
  size_t mem = 100;
  unsigned pipeline = 0;
  unsigned starts = 8;
  char bwt = 'a', enc = 'B';                                     // compress.cpp:115-118 defaults
  std::string prepr;                                             // compress.cpp:126-131: one 'p' per PairReplacer round
  bool from_stdin = false, to_stdout = false;
  int verbosity = 0, device = 0;
  std::vector<int> devices;
  static option longopts[] = {{"mem", required_argument, 0, 'm'},   {"starts", required_argument, 0, 's'},
                              {"bwt", required_argument, 0, 'b'},   {"enc", required_argument, 0, 'e'},
                              {"stdin", no_argument, 0, 'i'},       {"stdout", no_argument, 0, 'c'},
                              {"device", required_argument, 0, 'd'}, {"verb", required_argument, 0, 'v'},
                              {"devices", required_argument, 0, 'D'}, {"pipeline", required_argument, 0, 'P'},
                              {"prepr", required_argument, 0, 'p'},
                              {"help", no_argument, 0, 'h'},        {0, 0, 0, 0}};
  int o;
  while ((o = getopt_long(argc, argv, "m:s:e:icd:D:P:v:p:h", longopts, 0)) != -1) {
    switch (o) {
      case 'm': mem = std::strtoul(optarg, 0, 10); break;
      case 's': starts = (unsigned)std::strtoul(optarg, 0, 10); break;
      case 'b': bwt = optarg[0]; break;
      case 'e': enc = optarg[0]; break;
      case 'i': from_stdin = true; break;
      case 'c': to_stdout = true; break;
      case 'd': device = std::atoi(optarg); break;
      case 'D':
        for (const char* p = optarg; *p;) { devices.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        break;
      case 'P': pipeline = (unsigned)std::strtoul(optarg, 0, 10); break;
      case 'v': verbosity = std::atoi(optarg); break;
      case 'p':
        prepr = optarg;
        for (size_t i = 0; i < prepr.size(); ++i)
          if (prepr[i] != 'p') { std::fprintf(stderr, "Invalid choice for preprocessing: %c (p = pair replacer)\n", prepr[i]); return 1; }   // compress.cpp:45-60
        break;
      default:
        std::fprintf(stderr, "usage: compress [-m MB] [-s starts] [--bwt g] [-e B|b|u|H] [--prepr p...] [-i] [-c] [input] [output]\n");
        return o == 'h' ? 0 : 1;
    }
  }
  if (!bwtc::BWTManager::isValidChoice(bwt)) {                    // compress.cpp:86-96
    std::fprintf(stderr, "Invalid choice for BWT-algorithm: %c (this build offers g)\n", bwt);
    return 1;
  }
  if (mem < 1) mem = 1;
  std::string in_name, out_name;
  if (!from_stdin && optind < argc) in_name = argv[optind++];
  if (!to_stdout) {
    if (optind < argc) out_name = argv[optind++];
    else if (!in_name.empty()) out_name = in_name + ".bwtc";
  }
  if (!from_stdin && in_name.empty()) { std::fprintf(stderr, "no input\n"); return 1; }

  // One device and a stream from a pipe: the block farm with one context -- a reader thread and page-locked
  // staging, the upload of a block under the kernels of the block before it (160 blocks of 256 MiB from a
  // pipe: 12.6-13.3 s against 16.3 s for the loop below, which reads, uploads and transforms one after
  // the other).  Files keep the loop: read from the page cache it is the faster of the two (96 blocks:
  // 8.1 s against 8.8), and short streams pay for the farm's start-up.  BWTC_HIP_CLI_FARM=0 / 1 decides otherwise.
  if (devices.empty()) {
    bool farm = from_stdin;
    if (const char* f = std::getenv("BWTC_HIP_CLI_FARM")) farm = f[0] == '1';
    if (farm) devices.push_back(device);
  }

  // how many blocks a context keeps under way (the library reads it when the context is created)
  if (enc != 'H' && !std::getenv("BWTC_HIP_WAVELET_DEPTH")) {
    const size_t n_ctx = devices.empty() ? 1 : devices.size();
    if (pipeline == 0) {
      // Deep pipelines only for long streams: the fused engines keep every block under way for
      // seconds, which a stream of a few dozen blocks never earns back (24 blocks of 256 MiB:
      // 10.4 s instead of 7.6 s).
      const double gb = memAvailableGB() / n_ctx;
      const double block_bytes = mem * 0.185 * 1e6;
      double stream_bytes = 0.0;
      if (!from_stdin) {
        if (FILE* f = std::fopen(in_name.c_str(), "rb")) {
          if (fseeko(f, 0, SEEK_END) == 0) stream_bytes = (double)ftello(f);
          std::fclose(f);
        }
      }
      // ... and only where the adaptive models run on the worker threads (BWTC_HIP_MODELS=host): with
      // the models on the GPU -- the default, also for a stream farmed over several contexts -- a block
      // is under way for 0.7 s and holds 1.3 GB of page-locked memory; 13-14 blocks are what the rate
      // takes, 20 leave room for the spread of the host half (16: an occasional 2-3 ms wait per block).
      const char* mv = std::getenv("BWTC_HIP_MODELS");
      const bool host_models = (mv && std::strcmp(mv, "host") == 0) || enc != 'B';
      const bool deep = host_models && block_bytes >= 64e6 && stream_bytes / block_bytes >= 256.0 * n_ctx;
      pipeline = deep && gb >= 40.0 ? 128 : deep && gb >= 24.0 ? 96 : gb >= 48.0 ? 24 : gb >= 40.0 ? 20 : 16;   // an upper bound: the Compressor keeps what the stream shows to need (bwtc_hip_wavelet_depth_needed)
    }
    if (pipeline > 256) pipeline = 256;
    char buf[16];
    std::snprintf(buf, sizeof buf, "%u", pipeline);
    setenv("BWTC_HIP_WAVELET_DEPTH", buf, 1);
  } else if (enc != 'H') {                                       // the caller's environment sets the contexts' limit
    const unsigned lim = (unsigned)std::max(1, std::atoi(std::getenv("BWTC_HIP_WAVELET_DEPTH")));
    if (pipeline == 0 || pipeline > lim) pipeline = lim;
  }
  const auto t0 = std::chrono::steady_clock::now();
  bwtc::Compressor compressor(new bwtc::RawInStream(in_name), new bwtc::RawOutStream(out_name), prepr,
                              mem * 1000000, enc);                // compress.cpp:192-193
  size_t compressed;
  if (!prepr.empty() && !devices.empty()) {                       // the pre-stage runs block by block on one context
    std::fprintf(stderr, "compress: --prepr runs on one device; -D %s is ignored, device %d takes every block\n",
                 devices.size() > 1 ? "(the farm)" : "", device);
    devices.clear();
  }
  if (!devices.empty()) {
    compressed = compressor.compressFarmed(devices, starts < 1 ? 1 : starts > 256 ? 256 : starts, pipeline);
  } else {
    compressor.initializeBwtAlgorithm(bwt, starts, device);
    compressed = compressor.compress(1);
  }
  const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (verbosity > 0) std::fprintf(stderr, "Compressed size: %zu bytes, %.3f s\n", compressed, s);
  return 0;
}
