// dmt_pbrt_scene.cpp -- PBRT-v4 subset front-end: enough of the format for the reference's
// scenes/cornell-box.pbrt (:9-140), flattened to the megakernel's upload arrays.  The reference has NO loader for this
// file (it is rendered with pbrt itself; scenes/pbrt-output.png is pbrt's output), so directive semantics are pbrt-v4's:
//   Film "rgb" xresolution/yresolution      Sampler pixelsamples          LookAt eye look up
//   Camera "perspective" fov (shorter axis) WorldBegin                    AttributeBegin / AttributeEnd
//   Translate / Scale / Rotate (CTM = CTM * M)                            Identity, Transform / ConcatTransform (16 values)
//   MakeNamedMaterial name "string type" "diffuse" "rgb reflectance"      NamedMaterial name      Material "diffuse" ...
//   AreaLightSource "diffuse" "rgb L" [, "float scale"]                   Shape "trianglemesh" "point3 P" "integer indices"
//   LightSource "point" (I, from) | "spot" (I, from, to, coneangle, conedeltaangle) | "distant" (L, from, to) |
//               "infinite" (constant L): packed into the reference's own Light records (CC/private/light.cu:271-307)
//   Option / ColorSpace / Integrator / PixelFilter / Accelerator: accepted and ignored ("integer maxdepth" is read)
// Everything else is an error that names the directive.
//
// Mappings of this build: a pbrt "diffuse" material is Lambertian R/pi, which the reference's packed
// Oren-Nayar record with roughness 0 reproduces exactly; emissive shapes become dmt_upload_area_lights entries (one-sided,
// on the side of the pbrt normal).  pbrt's raster x axis runs against the megakernel camera's (for LookAt's handedness),
// so the scene is mirrored in the camera's right axis and every triangle rewound: the film then shows pbrt's picture.
// The camera must be the megakernel's kind: perspective, up = +z.  fov maps exactly for square films.
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include "dmt_scene.hpp"

namespace dmt_host {
namespace {

struct Fail {
  std::string msg;
};
[[noreturn]] void fail(std::string const& m) { throw Fail{m}; }

struct Token {
  enum Kind { Word, String, Number, LBracket, RBracket, End } kind = End;
  std::string text;
  double number = 0;
};

class Lexer {
 public:
  explicit Lexer(std::string const& s) : s_(s) {}
  Token next() {
    for (;;) {
      while (p_ < s_.size() && std::isspace(static_cast<unsigned char>(s_[p_]))) ++p_;
      if (p_ < s_.size() && s_[p_] == '#') {
        while (p_ < s_.size() && s_[p_] != '\n') ++p_;
        continue;
      }
      break;
    }
    Token t;
    if (p_ >= s_.size()) return t;
    char const c = s_[p_];
    if (c == '[') return ++p_, t.kind = Token::LBracket, t;
    if (c == ']') return ++p_, t.kind = Token::RBracket, t;
    if (c == '"') {
      size_t const e = s_.find('"', p_ + 1);
      if (e == std::string::npos) fail("unterminated string");
      t.kind = Token::String, t.text = s_.substr(p_ + 1, e - p_ - 1);
      p_ = e + 1;
      return t;
    }
    size_t e = p_;
    while (e < s_.size() && !std::isspace(static_cast<unsigned char>(s_[e])) && s_[e] != '[' && s_[e] != ']' && s_[e] != '"' && s_[e] != '#') ++e;
    t.text = s_.substr(p_, e - p_);
    p_ = e;
    char* endp = nullptr;
    double const v = std::strtod(t.text.c_str(), &endp);
    if (endp && *endp == 0 && !t.text.empty()) t.kind = Token::Number, t.number = v;
    else t.kind = Token::Word;
    return t;
  }

 private:
  std::string const& s_;
  size_t p_ = 0;
};

struct Param {
  std::string type, name;
  std::vector<double> nums;
  std::vector<std::string> strs;
};
using Params = std::vector<Param>;
Param const* find(Params const& ps, char const* type, char const* name) {
  for (auto const& p : ps)
    if (p.name == name && (p.type == type || (!strcmp(type, "rgb") && p.type == "color"))) return &p;
  return nullptr;
}

// pbrt's row-major 4x4 acting on column vectors
struct M4 {
  double m[4][4];
};
M4 identity() {
  M4 r{};
  for (int i = 0; i < 4; ++i) r.m[i][i] = 1;
  return r;
}
M4 mul(M4 const& a, M4 const& b) {
  M4 r{};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      for (int k = 0; k < 4; ++k) r.m[i][j] += a.m[i][k] * b.m[k][j];
  return r;
}
M4 translate(double x, double y, double z) {
  M4 r = identity();
  r.m[0][3] = x, r.m[1][3] = y, r.m[2][3] = z;
  return r;
}
M4 scale(double x, double y, double z) {
  M4 r = identity();
  r.m[0][0] = x, r.m[1][1] = y, r.m[2][2] = z;
  return r;
}
M4 rotate(double deg, double ax, double ay, double az) {  // pbrt Rotate(theta, axis)
  double const len = std::sqrt(ax * ax + ay * ay + az * az);
  if (!(len > 0)) fail("Rotate: zero axis");
  ax /= len, ay /= len, az /= len;
  double const t = deg * 3.14159265358979323846 / 180.0, s = std::sin(t), c = std::cos(t);
  M4 r = identity();
  r.m[0][0] = ax * ax + (1 - ax * ax) * c, r.m[0][1] = ax * ay * (1 - c) - az * s, r.m[0][2] = ax * az * (1 - c) + ay * s;
  r.m[1][0] = ax * ay * (1 - c) + az * s, r.m[1][1] = ay * ay + (1 - ay * ay) * c, r.m[1][2] = ay * az * (1 - c) - ax * s;
  r.m[2][0] = ax * az * (1 - c) - ay * s, r.m[2][1] = ay * az * (1 - c) + ax * s, r.m[2][2] = az * az + (1 - az * az) * c;
  return r;
}
Vec3 apply(M4 const& M, double x, double y, double z) {
  return Vec3{float(M.m[0][0] * x + M.m[0][1] * y + M.m[0][2] * z + M.m[0][3]), float(M.m[1][0] * x + M.m[1][1] * y + M.m[1][2] * z + M.m[1][3]),
              float(M.m[2][0] * x + M.m[2][1] * y + M.m[2][2] * z + M.m[2][3])};
}

struct GState {
  M4 ctm = identity();
  int material = -1;       // index into the material list, -1 = default
  bool emissive = false;
  float L[3] = {0, 0, 0};
};

}  // namespace

bool loadPbrtScene(std::string const& path, PbrtScene& out, std::string* error) {
  try {
    std::ifstream f(path);
    if (!f) fail("cannot open '" + path + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    std::string const text = ss.str();
    Lexer lex(text);
    Token tok = lex.next();
    auto advance = [&] { tok = lex.next(); };
    auto numbers = [&](int n, char const* what) {
      std::vector<double> v;
      for (int i = 0; i < n; ++i) {
        if (tok.kind != Token::Number) fail(std::string(what) + ": expected " + std::to_string(n) + " numbers");
        v.push_back(tok.number);
        advance();
      }
      return v;
    };
    auto stringArg = [&](char const* what) {
      if (tok.kind != Token::String) fail(std::string(what) + ": expected a quoted name");
      std::string s = tok.text;
      advance();
      return s;
    };
    auto params = [&] {  // "type name" value | [ values ]
      Params ps;
      while (tok.kind == Token::String) {
        Param p;
        std::istringstream decl(tok.text);
        if (!(decl >> p.type >> p.name)) fail("bad parameter declaration \"" + tok.text + "\"");
        advance();
        bool const bracket = tok.kind == Token::LBracket;
        if (bracket) advance();
        for (;;) {
          if (tok.kind == Token::Number) p.nums.push_back(tok.number);
          else if (tok.kind == Token::String) p.strs.push_back(tok.text);
          else if (tok.kind == Token::Word && (tok.text == "true" || tok.text == "false")) p.nums.push_back(tok.text == "true");
          else break;
          advance();
          if (!bracket) break;
        }
        if (bracket) {
          if (tok.kind != Token::RBracket) fail("parameter \"" + p.name + "\": missing ']'");
          advance();
        }
        ps.push_back(std::move(p));
      }
      return ps;
    };

    out = PbrtScene{};
    std::vector<GState> stack;
    GState gs;
    std::map<std::string, int> materialIndex;
    std::vector<Vec3> materials;  // reflectance
    auto addMaterial = [&](Params const& ps, std::string const& what) {
      std::string type = "diffuse";
      if (Param const* t = find(ps, "string", "type")) type = t->strs.empty() ? "" : t->strs[0];
      if (type != "diffuse") fail(what + ": only \"diffuse\" materials are supported, got \"" + type + "\"");
      Vec3 r{0.5f, 0.5f, 0.5f};
      if (Param const* p = find(ps, "rgb", "reflectance")) {
        if (p->nums.size() != 3) fail(what + ": \"rgb reflectance\" needs three values");
        r = Vec3{float(p->nums[0]), float(p->nums[1]), float(p->nums[2])};
      } else if (Param const* pf = find(ps, "float", "reflectance")) {
        if (pf->nums.size() != 1) fail(what + ": \"float reflectance\" needs one value");
        r = Vec3{float(pf->nums[0]), float(pf->nums[0]), float(pf->nums[0])};
      }
      materials.push_back(r);
      return int(materials.size()) - 1;
    };
    int defaultMaterial = -1;
    bool haveCamera = false, world = false;
    double fov = 90;
    Vec3 eye{0, 0, 0}, look{0, 1, 0}, up{0, 0, 1};
    struct PendingTri {
      Vec3 v[3];
      int material;
      bool emissive;
      float L[3];
    };
    std::vector<PendingTri> tris;
    struct PendingLight {
      int kind = 0;  // 0 point, 1 spot, 2 distant, 3 infinite (constant)
      Vec3 color, pos, to;
      float cos0 = 0.f, cosE = 0.f;
    };
    std::vector<PendingLight> lights;

    while (tok.kind != Token::End) {
      if (tok.kind != Token::Word) fail("expected a directive, got '" + tok.text + "'");
      std::string const d = tok.text;
      advance();
      if (d == "Film") {
        stringArg("Film");
        Params const ps = params();
        if (Param const* p = find(ps, "integer", "xresolution")) out.scene.camera.width = p->nums.empty() ? 0 : int(p->nums[0]);
        if (Param const* p = find(ps, "integer", "yresolution")) out.scene.camera.height = p->nums.empty() ? 0 : int(p->nums[0]);
      } else if (d == "Sampler") {
        stringArg("Sampler");
        Params const ps = params();
        if (Param const* p = find(ps, "integer", "pixelsamples")) out.samplesPerPixel = p->nums.empty() ? 1 : int(p->nums[0]);
      } else if (d == "Integrator") {
        stringArg("Integrator");
        Params const ps = params();
        if (Param const* p = find(ps, "integer", "maxdepth")) out.maxDepth = p->nums.empty() ? 5 : int(p->nums[0]);
      } else if (d == "Option" || d == "ColorSpace" || d == "PixelFilter" || d == "Accelerator") {
        if (d != "Option") stringArg(d.c_str());
        (void)params();
      } else if (d == "LookAt") {
        auto v = numbers(9, "LookAt");
        eye = Vec3{float(v[0]), float(v[1]), float(v[2])}, look = Vec3{float(v[3]), float(v[4]), float(v[5])}, up = Vec3{float(v[6]), float(v[7]), float(v[8])};
      } else if (d == "Camera") {
        std::string const type = stringArg("Camera");
        if (type != "perspective") fail("Camera \"" + type + "\": only \"perspective\" is supported");
        Params const ps = params();
        if (Param const* p = find(ps, "float", "fov")) fov = p->nums.empty() ? 90 : p->nums[0];
        haveCamera = true;
        gs.ctm = identity();  // pbrt: the CTM at Camera defines the camera; world space restarts at WorldBegin
      } else if (d == "WorldBegin") {
        world = true;
        gs = GState{};
      } else if (d == "AttributeBegin") {
        stack.push_back(gs);
      } else if (d == "AttributeEnd") {
        if (stack.empty()) fail("AttributeEnd without AttributeBegin");
        gs = stack.back();
        stack.pop_back();
      } else if (d == "Identity") {
        gs.ctm = identity();
      } else if (d == "Translate") {
        auto v = numbers(3, "Translate");
        gs.ctm = mul(gs.ctm, translate(v[0], v[1], v[2]));
      } else if (d == "Scale") {
        auto v = numbers(3, "Scale");
        gs.ctm = mul(gs.ctm, scale(v[0], v[1], v[2]));
      } else if (d == "Rotate") {
        auto v = numbers(4, "Rotate");
        gs.ctm = mul(gs.ctm, rotate(v[0], v[1], v[2], v[3]));
      } else if (d == "Transform" || d == "ConcatTransform") {
        if (tok.kind != Token::LBracket) fail(d + ": expected '['");
        advance();
        auto v = numbers(16, d.c_str());
        if (tok.kind != Token::RBracket) fail(d + ": expected ']'");
        advance();
        M4 m{};
        for (int i = 0; i < 16; ++i) m.m[i % 4][i / 4] = v[size_t(i)];  // pbrt files list matrices column by column
        gs.ctm = d == "Transform" ? m : mul(gs.ctm, m);
      } else if (d == "MakeNamedMaterial") {
        std::string const name = stringArg("MakeNamedMaterial");
        materialIndex[name] = addMaterial(params(), "MakeNamedMaterial \"" + name + "\"");
      } else if (d == "NamedMaterial") {
        std::string const name = stringArg("NamedMaterial");
        if (!materialIndex.count(name)) fail("NamedMaterial \"" + name + "\" is not defined");
        gs.material = materialIndex[name];
      } else if (d == "Material") {
        std::string const type = stringArg("Material");
        Params ps = params();
        Param t;
        t.type = "string", t.name = "type", t.strs = {type};
        ps.push_back(t);
        gs.material = addMaterial(ps, "Material");
      } else if (d == "AreaLightSource") {
        std::string const type = stringArg("AreaLightSource");
        if (type != "diffuse") fail("AreaLightSource \"" + type + "\": only \"diffuse\" is supported");
        Params const ps = params();
        Param const* L = find(ps, "rgb", "L");
        if (!L || L->nums.size() != 3) fail("AreaLightSource: \"rgb L\" with three values is required");
        double sc = 1;
        if (Param const* s = find(ps, "float", "scale")) sc = s->nums.empty() ? 1 : s->nums[0];
        if (Param const* two = find(ps, "bool", "twosided"))
          if (!two->nums.empty() && two->nums[0] != 0) fail("AreaLightSource: \"bool twosided\" true is not supported");
        gs.emissive = true;
        for (int i = 0; i < 3; ++i) gs.L[i] = float(L->nums[size_t(i)] * sc);
      } else if (d == "LightSource") {
        std::string const type = stringArg("LightSource");
        Params const ps = params();
        auto rgb = [&](char const* name, Vec3 dflt) {
          Param const* p = find(ps, "rgb", name);
          if (!p) return dflt;
          if (p->nums.size() != 3) fail(std::string("LightSource: \"rgb ") + name + "\" needs three values");
          return Vec3{float(p->nums[0]), float(p->nums[1]), float(p->nums[2])};
        };
        auto point = [&](char const* name, Vec3 dflt) {
          Param const* p = find(ps, "point3", name);
          if (!p) p = find(ps, "point", name);
          if (!p) return dflt;
          if (p->nums.size() != 3) fail(std::string("LightSource: \"point3 ") + name + "\" needs three values");
          return Vec3{float(p->nums[0]), float(p->nums[1]), float(p->nums[2])};
        };
        double sc = 1;
        if (Param const* sp = find(ps, "float", "scale")) sc = sp->nums.empty() ? 1 : sp->nums[0];
        auto scaled = [&](Vec3 v) { return Vec3{float(v.x * sc), float(v.y * sc), float(v.z * sc)}; };
        Vec3 const from = point("from", Vec3{0, 0, 0}), to = point("to", Vec3{0, 0, 1});
        Vec3 const wFrom = apply(gs.ctm, from.x, from.y, from.z), wTo = apply(gs.ctm, to.x, to.y, to.z);
        PendingLight pl;
        if (type == "point") {
          pl.kind = 0, pl.color = scaled(rgb("I", Vec3{1, 1, 1})), pl.pos = wFrom;
        } else if (type == "spot") {
          pl.kind = 1, pl.color = scaled(rgb("I", Vec3{1, 1, 1})), pl.pos = wFrom, pl.to = wTo;
          double cone = 30, delta = 5;
          if (Param const* c = find(ps, "float", "coneangle")) cone = c->nums.empty() ? 30 : c->nums[0];
          if (Param const* c = find(ps, "float", "conedeltaangle")) delta = c->nums.empty() ? 5 : c->nums[0];
          double const kRad = 3.14159265358979323846 / 180.0;
          pl.cos0 = float(std::cos((cone - delta) * kRad)), pl.cosE = float(std::cos(cone * kRad));  // falloff start, total width
        } else if (type == "distant") {
          pl.kind = 2, pl.color = scaled(rgb("L", Vec3{1, 1, 1})), pl.pos = wFrom, pl.to = wTo;
        } else if (type == "infinite") {
          if (find(ps, "string", "filename")) fail("LightSource \"infinite\": image maps are not supported (use the JSON front-end's envlight)");
          pl.kind = 3, pl.color = scaled(rgb("L", Vec3{1, 1, 1}));
        } else {
          fail("LightSource \"" + type + "\" is not supported");
        }
        lights.push_back(pl);
      } else if (d == "Shape") {
        if (!world) fail("Shape before WorldBegin");
        std::string const type = stringArg("Shape");
        if (type != "trianglemesh") fail("Shape \"" + type + "\": only \"trianglemesh\" is supported");
        Params const ps = params();
        Param const* P = find(ps, "point3", "P");
        if (!P) P = find(ps, "point", "P");
        if (!P || P->nums.size() % 3 || P->nums.empty()) fail("Shape \"trianglemesh\": \"point3 P\" is required");
        size_t const nv = P->nums.size() / 3;
        std::vector<int> idx;
        if (Param const* I = find(ps, "integer", "indices")) {
          for (double v : I->nums) idx.push_back(int(v));
        } else if (nv == 3) {
          idx = {0, 1, 2};
        } else {
          fail("Shape \"trianglemesh\": \"integer indices\" is required for more than three points");
        }
        if (idx.size() % 3) fail("Shape \"trianglemesh\": the number of indices is not a multiple of 3");
        if (gs.material < 0) {
          if (defaultMaterial < 0) defaultMaterial = addMaterial(Params{}, "default material");
        }
        for (size_t i = 0; i < idx.size(); i += 3) {
          PendingTri t;
          for (int k = 0; k < 3; ++k) {
            int const j = idx[i + size_t(k)];
            if (j < 0 || size_t(j) >= nv) fail("Shape \"trianglemesh\": index out of range");
            t.v[k] = apply(gs.ctm, P->nums[3 * size_t(j)], P->nums[3 * size_t(j) + 1], P->nums[3 * size_t(j) + 2]);
          }
          t.material = gs.material < 0 ? defaultMaterial : gs.material;
          t.emissive = gs.emissive;
          memcpy(t.L, gs.L, sizeof(t.L));
          tris.push_back(t);
        }
      } else {
        fail("directive '" + d + "' is not supported by this subset");
      }
    }
    if (!haveCamera) fail("no Camera directive");
    if (out.scene.camera.width <= 0 || out.scene.camera.height <= 0) fail("Film resolution is missing");
    if (std::fabs(up.x) > 1e-6f || std::fabs(up.y) > 1e-6f || !(up.z > 0.f)) fail("LookAt: only up = +z cameras are supported (the megakernel camera's convention)");

    // camera: DeviceCamera takes direction + position; fov (shorter axis) -> focal length for a 36 mm sensor
    Vec3 const dir{look.x - eye.x, look.y - eye.y, look.z - eye.z};
    out.scene.camera.dir[0] = dir.x, out.scene.camera.dir[1] = dir.y, out.scene.camera.dir[2] = dir.z;
    out.scene.camera.pos[0] = eye.x, out.scene.camera.pos[1] = eye.y, out.scene.camera.pos[2] = eye.z;
    out.scene.camera.spp = out.samplesPerPixel;
    out.scene.camera.sensor_size = 36.f;
    out.scene.camera.focal_length = float(18.0 / std::tan(fov * 3.14159265358979323846 / 360.0));

    // mirror the world in the camera's right axis (right = normalize(cross(dir, up)) is horizontal for up = +z) so that
    // the megakernel's raster x runs like pbrt's; a reflection flips orientation, so triangles are rewound
    float rx = dir.y * up.z - dir.z * up.y, ry = dir.z * up.x - dir.x * up.z, rz = dir.x * up.y - dir.y * up.x;
    float const rl = std::sqrt(rx * rx + ry * ry + rz * rz);
    if (!(rl > 0.f)) fail("LookAt: view direction is parallel to up");
    rx /= rl, ry /= rl, rz /= rl;
    auto mirror = [&](Vec3 p) {
      float const dist = (p.x - eye.x) * rx + (p.y - eye.y) * ry + (p.z - eye.z) * rz;
      return Vec3{p.x - 2.f * dist * rx, p.y - 2.f * dist * ry, p.z - 2.f * dist * rz};
    };
    for (PendingTri const& t : tris) {
      Vec3 const a = mirror(t.v[0]), b = mirror(t.v[2]), c = mirror(t.v[1]);
      out.scene.xs.insert(out.scene.xs.end(), {a.x, b.x, c.x, 0.f});
      out.scene.ys.insert(out.scene.ys.end(), {a.y, b.y, c.y, 0.f});
      out.scene.zs.insert(out.scene.zs.end(), {a.z, b.z, c.z, 0.f});
      out.scene.matId.push_back(uint32_t(t.material));
      if (t.emissive) {
        out.scene.areaTri.push_back(uint32_t(out.scene.matId.size() - 1));
        out.scene.areaLe.insert(out.scene.areaLe.end(), {t.L[0], t.L[1], t.L[2]});
      }
    }
    for (PendingLight const& l : lights) {  // mirrored like the geometry
      Vec3 const pos = mirror(l.pos), to = mirror(l.to);
      Vec3 dirv{to.x - pos.x, to.y - pos.y, to.z - pos.z};
      float const len = std::sqrt(dirv.x * dirv.x + dirv.y * dirv.y + dirv.z * dirv.z);
      if (l.kind == 1 || l.kind == 2) {
        if (!(len > 0.f)) fail("LightSource: \"from\" and \"to\" coincide");
        dirv = Vec3{dirv.x / len, dirv.y / len, dirv.z / len};
      }
      if (l.kind == 0) out.scene.lights.push_back(makePointLight(l.color, pos, 1e-3f));
      else if (l.kind == 1) out.scene.lights.push_back(makeSpotLight(l.color, pos, dirv, l.cos0, l.cosE, 1e-3f));
      else if (l.kind == 2) out.scene.lights.push_back(makeDirectionalLight(l.color, dirv, 0.f));
      else out.scene.infiniteLights.push_back(makeEnvironmentalLight(l.color));
    }
    for (Vec3 const& r : materials) out.scene.bsdfs.push_back(makeOrenNayar(r, 0.f));
    if (out.scene.bsdfs.empty()) out.scene.bsdfs.push_back(makeOrenNayar(Vec3{0.5f, 0.5f, 0.5f}, 0.f));
    return true;
  } catch (Fail const& e) {
    if (error) *error = e.msg;
    return false;
  } catch (std::exception const& e) {
    if (error) *error = e.what();
    return false;
  }
}

}  // namespace dmt_host
