// Drop-in for the reference's include/dataio.hpp (ghicp::DataIo<PointT>): the point-cloud file formats on either side of
// the registration path.  Host-only, no PCL: the PCD (v0.7 ascii / binary, the layouts pcl::io::savePCDFile* writes,
// including "_" padding fields and COUNT), PLY (ascii / binary_little_endian, vertex element) and TXT ("x y z" per line)
// readers and writers are self-contained so that clouds exchanged with the reference load on the GPU box.
//   readCloudFile / writeCloudFile   dataio.hpp:26-119 (dispatch on the extension)
//   readPcdFile / writePcdFile       dataio.hpp:121-139 (pcl::io::loadPCDFile / savePCDFileBinary)
//   readPlyFile / writePlyFile       dataio.hpp:490-506 (pcl::io::loadPLYFile / savePLYFile)
//   readTxtFile / writeTxtFile       dataio.hpp:508-585
//   outputKeypoints, savecoordinates dataio.hpp:587-627
// Not provided: the .las readers (libLAS) and their interactive global-shift prompts; binary_compressed PCD.
#ifndef GHICP_DROPIN_DATAIO_HPP_
#define GHICP_DROPIN_DATAIO_HPP_
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

#include "utility.h"

namespace ghicp {
namespace detail {
template <typename T, typename = void> struct has_intensity : std::false_type {};
template <typename T> struct has_intensity<T, decltype((void)std::declval<T&>().intensity, void())> : std::true_type {};
template <typename T> inline void set_intensity(T& p, float v, std::true_type) { p.intensity = v; }
template <typename T> inline void set_intensity(T&, float, std::false_type) {}
template <typename T> inline float get_intensity(const T& p, std::true_type) { return p.intensity; }
template <typename T> inline float get_intensity(const T&, std::false_type) { return 0.f; }

struct FieldDesc {
  std::string name;
  char type = 'F';  // F float, I signed, U unsigned
  int size = 4, count = 1, offset = 0;
};

inline double scalar_at(const unsigned char* p, char type, int size) {
  switch (type) {
    case 'F':
      if (size == 4) { float v; std::memcpy(&v, p, 4); return v; }
      if (size == 8) { double v; std::memcpy(&v, p, 8); return v; }
      break;
    case 'I':
      if (size == 1) { int8_t v; std::memcpy(&v, p, 1); return v; }
      if (size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; }
      if (size == 4) { int32_t v; std::memcpy(&v, p, 4); return v; }
      if (size == 8) { int64_t v; std::memcpy(&v, p, 8); return (double)v; }
      break;
    default:
      if (size == 1) { uint8_t v; std::memcpy(&v, p, 1); return v; }
      if (size == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; }
      if (size == 4) { uint32_t v; std::memcpy(&v, p, 4); return v; }
      if (size == 8) { uint64_t v; std::memcpy(&v, p, 8); return (double)v; }
  }
  return 0.0;
}
}  // namespace detail

template <typename PointT> class DataIo {
  typedef typename pcl::PointCloud<PointT>::Ptr CloudPtr;
  typedef detail::has_intensity<PointT> HasI;

 public:
  bool readCloudFile(const std::string& fileName, const CloudPtr& pointCloud) {
    const std::string ext = fileName.substr(fileName.find_last_of('.') + 1);
    bool ok = false;
    if (ext == "pcd") { ok = readPcdFile(fileName, pointCloud); if (ok) std::cout << "A pcd file has been imported" << std::endl; }
    else if (ext == "ply") { ok = readPlyFile(fileName, pointCloud); if (ok) std::cout << "A ply file has been imported" << std::endl; }
    else if (ext == "txt") { ok = readTxtFile(fileName, pointCloud); if (ok) std::cout << "A txt file has been imported" << std::endl; }
    else { std::cout << "Undefined Point Cloud Format." << std::endl; return false; }
    if (!ok) return false;
    std::cout << "Data loaded (" << pointCloud->points.size() << " points)" << std::endl;
    return true;
  }

  bool writeCloudFile(const std::string& fileName, const CloudPtr& pointCloud) {
    const std::string ext = fileName.substr(fileName.find_last_of('.') + 1);
    if (ext == "pcd") { if (!writePcdFile(fileName, pointCloud)) return false; std::cout << "A pcd file has been exported" << std::endl; }
    else if (ext == "ply") { if (!writePlyFile(fileName, pointCloud)) return false; std::cout << "A ply file has been exported" << std::endl; }
    else if (ext == "txt") { if (!writeTxtFile(fileName, pointCloud)) return false; std::cout << "A txt file has been exported" << std::endl; }
    else { std::cout << "Undefined Point Cloud Format." << std::endl; return false; }
    return true;
  }

  // ------------------------------------------------------------------------------------------------ PCD
  bool readPcdFile(const std::string& fileName, const CloudPtr& cloud) {
    std::ifstream in(fileName.c_str(), std::ios::binary);
    if (!in) return false;
    std::vector<detail::FieldDesc> f;
    size_t npts = 0, width = 0, height = 1;
    std::string data, line;
    while (std::getline(in, line)) {
      if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
      if (line.empty() || line[0] == '#') continue;
      std::istringstream ls(line);
      std::string key;
      ls >> key;
      if (key == "FIELDS") { std::string n; while (ls >> n) { detail::FieldDesc d; d.name = n; f.push_back(d); } }
      else if (key == "SIZE") { for (size_t i = 0; i < f.size(); i++) ls >> f[i].size; }
      else if (key == "TYPE") { for (size_t i = 0; i < f.size(); i++) ls >> f[i].type; }
      else if (key == "COUNT") { for (size_t i = 0; i < f.size(); i++) ls >> f[i].count; }
      else if (key == "WIDTH") ls >> width;
      else if (key == "HEIGHT") ls >> height;
      else if (key == "POINTS") ls >> npts;
      else if (key == "DATA") { ls >> data; break; }
    }
    if (f.empty() || data.empty()) return false;
    if (npts == 0) npts = width * height;
    int stride = 0, ix = -1, iy = -1, iz = -1, ii = -1;
    for (size_t i = 0; i < f.size(); i++) {
      f[i].offset = stride;
      stride += f[i].size * f[i].count;
      if (f[i].name == "x") ix = (int)i; else if (f[i].name == "y") iy = (int)i; else if (f[i].name == "z") iz = (int)i;
      else if (f[i].name == "intensity") ii = (int)i;
    }
    if (ix < 0 || iy < 0 || iz < 0) return false;
    cloud->points.clear();
    cloud->points.reserve(npts);
    if (data == "ascii") {
      for (size_t p = 0; p < npts; p++) {
        if (!std::getline(in, line)) break;
        std::istringstream ls(line);
        PointT pt = PointT();
        for (size_t i = 0; i < f.size(); i++)
          for (int c = 0; c < f[i].count; c++) {
            double v = 0;
            std::string tok;
            if (!(ls >> tok)) break;
            v = (tok == "nan" || tok == "NaN") ? std::numeric_limits<double>::quiet_NaN() : std::atof(tok.c_str());
            if (c == 0) assign(pt, (int)i, ix, iy, iz, ii, v);
          }
        cloud->points.push_back(pt);
      }
    } else if (data == "binary") {
      std::vector<unsigned char> rec((size_t)stride);
      for (size_t p = 0; p < npts; p++) {
        if (!in.read(reinterpret_cast<char*>(rec.data()), stride)) break;
        PointT pt = PointT();
        for (int i : {ix, iy, iz, ii})
          if (i >= 0) assign(pt, i, ix, iy, iz, ii, detail::scalar_at(rec.data() + f[i].offset, f[i].type, f[i].size));
        cloud->points.push_back(pt);
      }
    } else {
      return false;  // binary_compressed is not supported
    }
    finish(*cloud);
    return cloud->points.size() == npts;
  }

  // pcl::io::savePCDFileBinary: fields x y z (intensity), float32, tightly packed
  bool writePcdFile(const std::string& fileName, const CloudPtr& cloud) {
    std::ofstream out(fileName.c_str(), std::ios::binary);
    if (!out) return false;
    const size_t n = cloud->points.size();
    const bool wi = HasI::value;
    out << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z" << (wi ? " intensity" : "") << "\nSIZE 4 4 4" << (wi ? " 4" : "")
        << "\nTYPE F F F" << (wi ? " F" : "") << "\nCOUNT 1 1 1" << (wi ? " 1" : "") << "\nWIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n
        << "\nDATA binary\n";
    for (size_t i = 0; i < n; i++) {
      const PointT& p = cloud->points[i];
      const float v[4] = {p.x, p.y, p.z, detail::get_intensity(p, HasI())};
      out.write(reinterpret_cast<const char*>(v), wi ? 16 : 12);
    }
    return (bool)out;
  }

  // ------------------------------------------------------------------------------------------------ PLY
  bool readPlyFile(const std::string& fileName, const CloudPtr& cloud) {
    std::ifstream in(fileName.c_str(), std::ios::binary);
    if (!in) return false;
    std::string line, format;
    if (!std::getline(in, line) || line.substr(0, 3) != "ply") return false;
    std::vector<detail::FieldDesc> f;
    size_t nvert = 0;
    bool in_vertex = false, vertex_first = true, seen_element = false;
    while (std::getline(in, line)) {
      if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
      std::istringstream ls(line);
      std::string key;
      ls >> key;
      if (key == "format") ls >> format;
      else if (key == "element") {
        std::string name;
        size_t cnt;
        ls >> name >> cnt;
        in_vertex = name == "vertex";
        if (in_vertex) { nvert = cnt; vertex_first = !seen_element; }
        seen_element = true;
      } else if (key == "property" && in_vertex) {
        std::string ty, name;
        ls >> ty;
        if (ty == "list") return false;  // no list properties on vertices
        ls >> name;
        detail::FieldDesc d;
        d.name = name;
        if (ty == "float" || ty == "float32") { d.type = 'F'; d.size = 4; }
        else if (ty == "double" || ty == "float64") { d.type = 'F'; d.size = 8; }
        else if (ty == "char" || ty == "int8") { d.type = 'I'; d.size = 1; }
        else if (ty == "uchar" || ty == "uint8") { d.type = 'U'; d.size = 1; }
        else if (ty == "short" || ty == "int16") { d.type = 'I'; d.size = 2; }
        else if (ty == "ushort" || ty == "uint16") { d.type = 'U'; d.size = 2; }
        else if (ty == "int" || ty == "int32") { d.type = 'I'; d.size = 4; }
        else if (ty == "uint" || ty == "uint32") { d.type = 'U'; d.size = 4; }
        else return false;
        f.push_back(d);
      } else if (key == "end_header") break;
    }
    if (!vertex_first || f.empty()) return false;  // the vertex element must come first (it does in every writer we exchange with)
    int stride = 0, ix = -1, iy = -1, iz = -1, ii = -1;
    for (size_t i = 0; i < f.size(); i++) {
      f[i].offset = stride;
      stride += f[i].size;
      if (f[i].name == "x") ix = (int)i; else if (f[i].name == "y") iy = (int)i; else if (f[i].name == "z") iz = (int)i;
      else if (f[i].name == "intensity" || f[i].name == "scalar_intensity") ii = (int)i;
    }
    if (ix < 0 || iy < 0 || iz < 0) return false;
    cloud->points.clear();
    cloud->points.reserve(nvert);
    if (format == "ascii") {
      for (size_t p = 0; p < nvert; p++) {
        if (!std::getline(in, line)) break;
        std::istringstream ls(line);
        PointT pt = PointT();
        for (size_t i = 0; i < f.size(); i++) {
          double v;
          if (!(ls >> v)) break;
          assign(pt, (int)i, ix, iy, iz, ii, v);
        }
        cloud->points.push_back(pt);
      }
    } else if (format == "binary_little_endian") {
      std::vector<unsigned char> rec((size_t)stride);
      for (size_t p = 0; p < nvert; p++) {
        if (!in.read(reinterpret_cast<char*>(rec.data()), stride)) break;
        PointT pt = PointT();
        for (int i : {ix, iy, iz, ii})
          if (i >= 0) assign(pt, i, ix, iy, iz, ii, detail::scalar_at(rec.data() + f[i].offset, f[i].type, f[i].size));
        cloud->points.push_back(pt);
      }
    } else {
      return false;
    }
    finish(*cloud);
    return cloud->points.size() == nvert;
  }

  // pcl::io::savePLYFile default: ascii
  bool writePlyFile(const std::string& fileName, const CloudPtr& cloud) {
    std::ofstream out(fileName.c_str());
    if (!out) return false;
    const size_t n = cloud->points.size();
    out << "ply\nformat ascii 1.0\ncomment ghicp-hip generated\nelement vertex " << n << "\nproperty float x\nproperty float y\nproperty float z\n";
    if (HasI::value) out << "property float intensity\n";
    out << "end_header\n";
    out << std::setprecision(9);
    for (size_t i = 0; i < n; i++) {
      const PointT& p = cloud->points[i];
      out << p.x << " " << p.y << " " << p.z;
      if (HasI::value) out << " " << detail::get_intensity(p, HasI());
      out << "\n";
    }
    return (bool)out;
  }

  // ------------------------------------------------------------------------------------------------ TXT
  bool readTxtFile(const std::string& fileName, const CloudPtr& pointCloud) {  // dataio.hpp:508-534: "x y z" as doubles until the stream fails
    std::ifstream in(fileName.c_str(), std::ios::in);
    if (!in) return false;
    double x_ = 0, y_ = 0, z_ = 0;
    while (!in.eof()) {
      in >> x_ >> y_ >> z_;
      if (in.fail()) break;
      PointT pt = PointT();
      pt.x = (float)x_; pt.y = (float)y_; pt.z = (float)z_;
      pointCloud->points.push_back(pt);
    }
    finish(*pointCloud);
    return true;
  }

  bool writeTxtFile(const std::string& fileName, const CloudPtr& pointCloud) { return writeTxtFile(fileName, pointCloud, 1); }

  bool writeTxtFile(const std::string& fileName, const CloudPtr& pointCloud, int subsample_ratio) {  // dataio.hpp:536-585
    std::ofstream ofs(fileName.c_str());
    if (!ofs.is_open()) return false;
    for (size_t i = 0; i < pointCloud->points.size(); ++i)
      if (subsample_ratio <= 1 || i % (size_t)subsample_ratio == 0)
        ofs << std::setiosflags(std::ios::fixed) << std::setprecision(6) << pointCloud->points[i].x << "  " << pointCloud->points[i].y << "  "
            << pointCloud->points[i].z << std::endl;
    return true;
  }

  bool outputKeypoints(const std::string& filename, const pcl::PointIndicesPtr& indices, const CloudPtr& pointCloud) {  // dataio.hpp:587-607
    std::ofstream ofs(filename.c_str());
    if (!ofs.is_open()) return false;
    for (size_t i = 0; i < indices->indices.size(); ++i) {
      const PointT& p = pointCloud->points[(size_t)indices->indices[i]];
      ofs << std::setiosflags(std::ios::fixed) << std::setprecision(6) << p.x << "\t" << p.y << "\t" << p.z << std::endl;
    }
    return true;
  }

  bool savecoordinates(const CloudPtr& Source_FPC, const CloudPtr& Target_FPC, pcl::PointIndicesPtr& Source_KPI, pcl::PointIndicesPtr& Target_KPI,
                       Eigen::MatrixX3d& SXYZ, Eigen::MatrixX3d& TXYZ) {  // dataio.hpp:609-627: keypoint xyz (f32) widened to f64
    SXYZ.resize((long)Source_KPI->indices.size(), 3);
    TXYZ.resize((long)Target_KPI->indices.size(), 3);
    for (size_t i = 0; i < Source_KPI->indices.size(); ++i) {
      const PointT& p = Source_FPC->points[(size_t)Source_KPI->indices[i]];
      SXYZ((long)i, 0) = p.x; SXYZ((long)i, 1) = p.y; SXYZ((long)i, 2) = p.z;
    }
    for (size_t i = 0; i < Target_KPI->indices.size(); ++i) {
      const PointT& p = Target_FPC->points[(size_t)Target_KPI->indices[i]];
      TXYZ((long)i, 0) = p.x; TXYZ((long)i, 1) = p.y; TXYZ((long)i, 2) = p.z;
    }
    std::cout << "Key points saved." << std::endl;
    return true;
  }

 private:
  static void assign(PointT& pt, int i, int ix, int iy, int iz, int ii, double v) {
    if (i == ix) pt.x = (float)v;
    else if (i == iy) pt.y = (float)v;
    else if (i == iz) pt.z = (float)v;
    else if (i == ii) detail::set_intensity(pt, (float)v, HasI());
  }
  static void finish(pcl::PointCloud<PointT>& c) {
    c.width = (unsigned)c.points.size();
    c.height = 1;
  }
};
}  // namespace ghicp
#endif
