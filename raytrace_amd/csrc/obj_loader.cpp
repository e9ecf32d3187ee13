// obj_loader.cpp -- OBJ subset + the reference's MTL dialect.
//
// Behaviour restated from core/ObjLoader.cpp:12-34 (tokenizer), :37-224
// (LoadObj), :226-242 (ResetMaterial), :244-354 (LoadObjMaterials) and
// core/MtlLib.cpp:3-12.  Record grammar:
//   v x y z            -> vertex (x,y,z,1); fewer than 4 tokens -> (0,0,0,1)
//   vn x y z           -> normal (x,y,z,0); fewer than 4 tokens -> (1,0,0,0)
//   vt u v [w]         -> texture coordinate; kept, with the per-corner indices the reference leaves commented out
//   f a b c            -> exactly 4 tokens, each "v", "v/vt" or "v/vt/vn"; 1-based
//   mtllib file        -> load materials from folder + file
//   usemtl name        -> current material index (unknown name -> 0)
// In kLoaderReference mode a run of k spaces yields k-1 empty tokens, which is
// what makes "v  x y z" load as (0,x,y) and "f  1  7  5" count 7 tokens.
#include "host_scene.h"

#include <cstdlib>
#include <fstream>

namespace rt {

void MtlLib::InsertMaterial(std::string const &name, rt_material const &mtl)
{
  indices[name] = static_cast<int>(materials.size());
  materials.push_back(mtl);
}

int MtlLib::MaterialIndex(std::string const &name) { return indices[name]; }

void SplitString(std::string const &str, char token, std::vector<std::string> &out, bool keep_empty)
{
  out.clear();
  size_t begin = 0;
  for (;;) {
    size_t at = str.find(token, begin);
    if (at == std::string::npos) break;
    if (at != begin)
      out.emplace_back(str, begin, at - begin);
    else if (keep_empty)
      out.emplace_back();
    begin = at + 1;
  }
  if (begin < str.size()) out.emplace_back(str, begin, std::string::npos);
}

namespace {

inline float ToFloat(std::string const &s) { return static_cast<float>(std::atof(s.c_str())); }

struct LineReader {
  std::istream &in;
  std::string line, type;
  std::vector<std::string> tok;
  bool keep_empty;
  // Returns false at end of stream.  Empty lines come back with type "".
  bool Next()
  {
    if (!in) return false;
    line.clear();  // a getline that fails at end of file leaves its argument untouched
    std::getline(in, line);
    type = line.substr(0, line.find_first_of(' '));
    return true;
  }
  void Tokens() { SplitString(line, ' ', tok, keep_empty); }
};

void ResetMaterial(rt_material &m)
{
  m.color = rt_float4{0, 0, 0, 0};
  m.uv = rt_float4{0, 0, 0, 0};
  m.Si = 1;
  m.Rf = 1;
  m.texId = 0;
}

// One "a/b/c" face corner: field 0 = vertex, field 2 = normal; '/'-splitting
// always keeps empty fields ("1//3").
struct Corner {
  std::vector<std::string> f;
  explicit Corner(std::string const &s) { SplitString(s, '/', f, true); }
  bool HasVertex() const { return !f.empty() && !f[0].empty(); }
  bool HasNormal() const { return f.size() > 2 && !f[2].empty(); }
  bool HasTexcoord() const { return f.size() > 1 && !f[1].empty(); }
};

}  // namespace

void LoadObjMaterials(std::istream &file, MtlLib &matlib, LoaderMode mode, std::vector<std::string> *log)
{
  if (!file) return;
  rt_material mat{};
  std::string name;
  LineReader rd{file, {}, {}, {}, mode == kLoaderReference};
  while (rd.Next()) {
    if (rd.line.empty()) continue;
    if (rd.type == "newmtl") {
      if (!name.empty()) {
        if (matlib.indices.find(name) != matlib.indices.end()) {
          if (log) log->push_back("Material already defined, using previous definition");
        } else {
          matlib.InsertMaterial(name, mat);
        }
      }
      ResetMaterial(mat);
      rd.Tokens();
      name = rd.tok.size() > 1 ? rd.tok[1] : std::string();
    } else if (rd.type == "color") {
      rd.Tokens();
      if (rd.tok.size() > 3) {
        mat.color.x = ToFloat(rd.tok[1]);
        mat.color.y = ToFloat(rd.tok[2]);
        mat.color.z = ToFloat(rd.tok[3]);
      }
    } else if (rd.type == "rough") {
      rd.Tokens();
      if (rd.tok.size() > 1) mat.Rf = ToFloat(rd.tok[1]);
    } else if (rd.type == "ind_ref") {
      rd.Tokens();
      if (rd.tok.size() > 1) mat.Si = ToFloat(rd.tok[1]);
    }
  }
  // The trailing definition is inserted without the duplicate check (core/ObjLoader.cpp:348-352).
  if (!name.empty()) matlib.InsertMaterial(name, mat);
}

void LoadObj(std::istream &file, Mesh &mesh, std::string const &folder, LoaderMode mode)
{
  if (!file) {
    mesh.log.push_back("stream empty");
    return;
  }
  std::vector<rt_vec3> vertices;
  std::vector<rt_float4> normals;
  std::vector<rt_face> faces;
  std::vector<rt_vec3> uvws;
  std::vector<uint32_t> face_uvws;
  int curr_mat_index = 0;
  LineReader rd{file, {}, {}, {}, mode == kLoaderReference};

  while (rd.Next()) {
    if (rd.line.empty()) continue;
    std::string const &ty = rd.type;
    if (ty == "v") {
      rd.Tokens();
      rt_vec3 v{0, 0, 0, 1};
      if (rd.tok.size() < 4) {
        mesh.log.push_back("incomplete vertex record, using (0,0,0)");
      } else {
        v.x = ToFloat(rd.tok[1]);
        v.y = ToFloat(rd.tok[2]);
        v.z = ToFloat(rd.tok[3]);
      }
      vertices.push_back(v);
    } else if (ty == "vn") {
      rd.Tokens();
      rt_float4 n{1, 0, 0, 0};
      if (rd.tok.size() < 4) {
        mesh.log.push_back("incomplete texture coordinate record, using <1,0,0>");
      } else {
        n.x = ToFloat(rd.tok[1]);
        n.y = ToFloat(rd.tok[2]);
        n.z = ToFloat(rd.tok[3]);
      }
      normals.push_back(n);
    } else if (ty == "vt") {
      // core/ObjLoader.cpp:96-118: fewer than two values -> (0,0); a third value is optional (the reference leaves
      // z unset without one; 0 here)
      rd.Tokens();
      rt_vec3 uvw{0, 0, 0, 0};
      if (rd.tok.size() < 3) {
        mesh.log.push_back("incomplete texture coordinate record, using (0,0)");
      } else {
        uvw.x = ToFloat(rd.tok[1]);
        uvw.y = ToFloat(rd.tok[2]);
        if (rd.tok.size() > 3) uvw.z = ToFloat(rd.tok[3]);
      }
      uvws.push_back(uvw);
    } else if (ty == "f") {
      rd.Tokens();
      if (rd.tok.size() < 4) {
        mesh.log.push_back("incomplete face record, ignoring");
      } else if (rd.tok.size() > 4 && mode != kLoaderExtended) {
        mesh.log.push_back("Quads and ngons not supported, ignoring.");
      } else {
        // One triangle in the reference and tolerant modes; a fan (c0, c_i, c_i+1) over quads and n-gons in the
        // extended mode, which also resolves the negative, count-relative indices OBJ allows.
        auto index = [&](std::string const &field, size_t count) -> unsigned {
          const long v = std::atol(field.c_str());
          if (mode == kLoaderExtended && v < 0) return static_cast<unsigned>(static_cast<long>(count) + v);
          return static_cast<unsigned>(v) - 1;
        };
        Corner c0(rd.tok[1]);
        for (size_t k = 2; k + 1 < rd.tok.size(); ++k) {
          Corner c1(rd.tok[k]), c2(rd.tok[k + 1]);
          if (!c0.HasVertex() || !c1.HasVertex() || !c2.HasVertex()) {
            mesh.log.push_back("face record missing vertex index, skipping");
            break;
          }
          rt_face face{};
          face.s[0] = index(c0.f[0], vertices.size());
          face.s[1] = index(c1.f[0], vertices.size());
          face.s[2] = index(c2.f[0], vertices.size());
          face.s[3] = static_cast<unsigned>(curr_mat_index);
          if (c0.HasNormal() && c1.HasNormal() && c2.HasNormal()) {
            face.s[4] = index(c0.f[2], normals.size());
            face.s[5] = index(c1.f[2], normals.size());
            face.s[6] = index(c2.f[2], normals.size());
          }
          faces.push_back(face);
          const bool has_uv = c0.HasTexcoord() && c1.HasTexcoord() && c2.HasTexcoord();
          face_uvws.push_back(has_uv ? index(c0.f[1], uvws.size()) : 0xFFFFFFFFu);
          face_uvws.push_back(has_uv ? index(c1.f[1], uvws.size()) : 0xFFFFFFFFu);
          face_uvws.push_back(has_uv ? index(c2.f[1], uvws.size()) : 0xFFFFFFFFu);
        }
      }
    } else if (ty == "mtllib") {
      rd.Tokens();
      if (rd.tok.size() > 1) {
        std::ifstream mtl(folder + rd.tok[1]);
        LoadObjMaterials(mtl, mesh.mtllib, mode, &mesh.log);  // a missing file is silently ignored
      }
    } else if (ty == "usemtl") {
      rd.Tokens();
      curr_mat_index = mesh.mtllib.MaterialIndex(rd.tok.size() > 1 ? rd.tok[1] : std::string());
    }
    // '#' and every other record type carry nothing this path uses.
  }
  mesh.verts = std::move(vertices);
  mesh.faces = std::move(faces);
  mesh.normals = std::move(normals);
  mesh.uvws = std::move(uvws);
  mesh.face_uvws = std::move(face_uvws);
}

bool LoadObjFile(std::string const &path, Mesh &mesh, LoaderMode mode)
{
  std::ifstream in(path);
  if (!in) return false;
  std::string folder;
  size_t slash = path.find_last_of('/');
  if (slash != std::string::npos) folder = path.substr(0, slash + 1);
  rt_material sky{};
  mesh.mtllib.InsertMaterial("SKY_DEFAULT", sky);
  LoadObj(in, mesh, folder, mode);
  return true;
}

}  // namespace rt
