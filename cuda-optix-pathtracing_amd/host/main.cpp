// dmt-megakernel-hip -- command-line driver with the reference's dmt-megakernel surface
// (examples/triangles/megakernel/main.cu:67-243; flags CC/private/host_utils.cu:39-92):
//   --width <N> --height <N> --spp <N> --kspp <N> --log-level info|verbose --save-partial
// plus --max-depth <N> (reference constant 32), --device <ordinal>, --out <dir>, --scene <file.json>, --bvh.
// Renders the hard-coded cornellBox() scene kspp samples per launch and writes
// output-<spp>.png and output-<spp>_sqrt_mse.png next to the executable (or into --out).
#include <unistd.h>

#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dmt_scene.hpp"

namespace {

struct Config {  // defaults: CC/public/cuda-core/host_utils.cuh:25-31
  int width = 256, height = 256, spp = 2048, kspp = 4;
  int maxDepth = 32, device = 0;
  std::string logLevel = "info", outDir;
  bool savePartial = false;
  std::string scenePath;  // --scene <file.json>: the reference's JSON scene description instead of cornellBox()
  bool bvh = false;       // --bvh: traverse the 4-wide BVH instead of testing every triangle
  bool widthSet = false, heightSet = false, sppSet = false, depthSet = false;

  std::string validate() const {  // host_utils.cuh:35-62
    if (width <= 0) return "invalid width: should be bigger than zero. got " + std::to_string(width);
    if (height <= 0) return "invalid height: should be bigger than zero. got " + std::to_string(height);
    if (spp <= 0) return "invalid spp: should be bigger than zero. got " + std::to_string(spp);
    if (spp < kspp) return "invalid spp: should be bigger than kspp. got " + std::to_string(spp) + " and kspp" + std::to_string(kspp);
    if (kspp <= 0) return "invalid kspp: should be bigger than zero. got " + std::to_string(kspp);
    if (logLevel != "info" && logLevel != "verbose") return "invalid logLevel value. Either info or verbose, got " + logLevel;
    if (maxDepth < 0) return "invalid max-depth";
    return "";
  }
};

void printHelp() {
  std::puts(
      "Input Commands:\n"
      "  --width <N>       -- Define Width of output image\n"
      "  --height <N>      -- Define Height of output image\n"
      "  --spp <N>         -- Define Samples per pixel\n"
      "  --kspp <N>        -- Define Samples per pixel processed on a single kernel loop\n"
      "  --log-level <N>   -- Log Verbosity, 'info' or 'verbose'\n"
      "  --save-partial    -- Whether to save images every <kspp> samples\n"
      "  --max-depth <N>   -- Bounce cap (reference: 32)\n"
      "  --device <N>      -- GPU ordinal\n"
      "  --out <dir>       -- Output directory (default: the executable's directory)\n"
      "  --scene <file>    -- JSON scene (camera/film/materials/objects/lights/envlight/transforms/world) or *.pbrt\n"
      "                       (PBRT-v4 subset: diffuse materials, triangle meshes, diffuse area lights);\n"
      "                       its resolution, samples and max-depth apply unless given on the command line\n"
      "  --bvh             -- BVH traversal instead of the brute-force triangle loop");
}

Config parseArguments(int argc, char** argv) {
  Config c;
  for (int i = 1; i < argc; ++i) {
    std::string const a = argv[i];
    bool const more = i + 1 < argc;
    if (a == "--width" && more) c.width = std::atoi(argv[++i]), c.widthSet = true;
    else if (a == "--height" && more) c.height = std::atoi(argv[++i]), c.heightSet = true;
    else if (a == "--spp" && more) c.spp = std::atoi(argv[++i]), c.sppSet = true;
    else if (a == "--scene" && more) c.scenePath = argv[++i];
    else if (a == "--bvh") c.bvh = true;
    else if (a == "--kspp" && more) c.kspp = std::atoi(argv[++i]);
    else if (a == "--log-level" && more) c.logLevel = argv[++i];
    else if (a == "--save-partial") c.savePartial = true;
    else if (a == "--max-depth" && more) c.maxDepth = std::atoi(argv[++i]), c.depthSet = true;
    else if (a == "--device" && more) c.device = std::atoi(argv[++i]);
    else if (a == "--out" && more) c.outDir = argv[++i];
    else if (a == "--help") { printHelp(); std::exit(0); }
  }
  return c;
}

std::string executableDirectory() {
  char buf[PATH_MAX];
  ssize_t const n = readlink("/proc/self/exe", buf, sizeof(buf) - 1);
  if (n <= 0) return ".";
  buf[n] = 0;
  std::string p(buf);
  size_t const slash = p.find_last_of('/');
  return slash == std::string::npos ? "." : p.substr(0, slash);
}

int fail(dmt_ctx* ctx, char const* what) {
  std::fprintf(stderr, "%s failed: %s\n", what, dmt_last_error(ctx));
  if (ctx) dmt_ctx_destroy(ctx);
  return 1;
}

}  // namespace

int main(int argc, char** argv) {
  Config cfg = parseArguments(argc, argv);
  dmt_host::JsonScene json;
  if (!cfg.scenePath.empty()) {
    std::string err;
    bool const pbrt = cfg.scenePath.size() > 5 && cfg.scenePath.compare(cfg.scenePath.size() - 5, 5, ".pbrt") == 0;
    bool ok;
    if (pbrt) {  // PBRT-v4 subset (scenes/cornell-box.pbrt); same fields as the JSON front-end
      dmt_host::PbrtScene ps;
      ok = dmt_host::loadPbrtScene(cfg.scenePath, ps, &err);
      json.scene = std::move(ps.scene), json.maxDepth = ps.maxDepth, json.samplesPerPixel = ps.samplesPerPixel;
    } else {
      ok = dmt_host::loadJsonScene(cfg.scenePath, json, &err);
    }
    if (!ok) {
      std::fprintf(stderr, "scene '%s': %s\n", cfg.scenePath.c_str(), err.c_str());
      return 1;
    }
    if (!cfg.widthSet) cfg.width = json.scene.camera.width;
    if (!cfg.heightSet) cfg.height = json.scene.camera.height;
    if (!cfg.sppSet) cfg.spp = json.samplesPerPixel;
    if (!cfg.depthSet) cfg.maxDepth = json.maxDepth;
    if (cfg.kspp > cfg.spp) cfg.kspp = cfg.spp;
  }
  if (std::string const err = cfg.validate(); !err.empty()) {
    std::fprintf(stderr, "%s\n", err.c_str());
    printHelp();
    return 1;
  }
  std::printf("Parsed Configuration:\n - Width:     %d\n - Height:    %d\n - SPP:       %d\n - KSPP:      %d\n - Log Level: %s\n",
              cfg.width, cfg.height, cfg.spp, cfg.kspp, cfg.logLevel.c_str());
  bool const verbose = cfg.logLevel == "verbose";

  dmt_ctx* ctx = nullptr;
  if (dmt_ctx_create(cfg.device, &ctx) != DMT_OK) return fail(nullptr, "dmt_ctx_create");
  dmt_host::Scene scene = cfg.scenePath.empty() ? dmt_host::cornellBox() : std::move(json.scene);
  scene.camera.width = cfg.width, scene.camera.height = cfg.height, scene.camera.spp = cfg.kspp;
  if (dmt_host::uploadScene(ctx, scene) != DMT_OK) return fail(ctx, "uploadScene");
  if (dmt_set_limits(ctx, cfg.maxDepth) != DMT_OK) return fail(ctx, "dmt_set_limits");
  if (cfg.bvh && dmt_set_accel(ctx, DMT_ACCEL_BVH) != DMT_OK) return fail(ctx, "dmt_set_accel");

  std::string const dir = cfg.outDir.empty() ? executableDirectory() : cfg.outDir;
  size_t const pixels = size_t(cfg.width) * size_t(cfg.height);
  std::vector<float> mean(4 * pixels), m2(4 * pixels);
  auto writeOut = [&](int samples) {
    if (dmt_download_film(ctx, mean.data(), m2.data()) != DMT_OK) return false;
    std::string err;
    std::puts("Writing to file");
    if (!dmt_host::writeMeanAndMSERowMajor(mean.data(), m2.data(), uint32_t(cfg.width), uint32_t(cfg.height),
                                           dir + "/output-" + std::to_string(samples), &err)) {
      std::fprintf(stderr, "%s\n", err.c_str());
      return false;
    }
    return true;
  };

  std::puts("Running HIP Kernel");
  double totalMs = 0.0;  // wall time of launch + sync, file writes excluded (main.cu:179-192)
  int launches = 0;
  for (int sTot = 0; sTot < cfg.spp; sTot += cfg.kspp) {
    if (verbose) std::printf("Running HIP Kernel (%d)\n", sTot);
    auto const t0 = std::chrono::steady_clock::now();
    if (dmt_render(ctx, uint32_t(sTot), uint32_t(cfg.kspp), 0, 0, cfg.width, cfg.height) != DMT_OK) return fail(ctx, "dmt_render");
    if (dmt_sync(ctx) != DMT_OK) return fail(ctx, "dmt_sync");
    totalMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ++launches;
    if (cfg.savePartial && !writeOut(sTot + cfg.kspp)) return fail(ctx, "write");
  }
  if (!cfg.savePartial && !writeOut(cfg.spp)) return fail(ctx, "write");
  double const samples = double(pixels) * double(launches) * double(cfg.kspp);
  std::printf("Done! Total Execution Time(excl write file): %llu ms | Average Execution per Kernel launch (%d spp): %llu ms | %.2f Msamples/s\n",
              static_cast<unsigned long long>(totalMs), cfg.kspp, static_cast<unsigned long long>(totalMs / launches),
              samples / (totalMs * 1e3));
  std::puts("Cleanup...");
  dmt_ctx_destroy(ctx);
  return 0;
}
