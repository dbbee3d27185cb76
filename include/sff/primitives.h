// primitives.h — source-compatible subset of the reference's src/primitives.h for the drop-in
// header set (include/sff/).  Only what src/main.cpp and the solver front ends touch is provided:
// Point / Range / Dimensions / FileStruct / Node statics / helpers.  The hot-path geometry itself
// lives in libsffgpu (csrc/sff_geom.h); these types are the host-side carriers.
#pragma once
#include <cmath>
#include <deque>
#include <map>
#include <regex>
#include <stdexcept>
#include <string>
#include <vector>

#ifndef DELIMITER_OUT
#define DELIMITER_OUT (" ")
#endif
#define CSV_DELIMITER (",")
#define CSV_DELIMITER_2 (";")
#define TSP_DELIMITER (" ")
#define TOLERANCE 1e-9
#define DEFAULT_THRES_MISS 3
#define DEFAULT_SAMP_DIST 1

enum Dimensions { D2 = 2, D3 = 6 };
enum FileType { Map, Obj };
struct FileStruct {
  std::string fileName;
  FileType type;
};

template <class T>
struct Range {
  T minX, maxX, minY, maxY, minZ, maxZ;
};

// reference src/primitives.h:86-275 (public surface used by the parser and the writers)
template <class T>
class Point {
 public:
  T Yaw{0}, Pitch{0}, Roll{0};
  Point() : c{0, 0, 0} {}
  Point(T x, T y, T z) : c{x, y, z} {}
  Point(T x, T y, T z, T yaw, T pitch, T roll) : Yaw{yaw}, Pitch{pitch}, Roll{roll}, c{x, y, z} {}
  // "[x; y; z]" scaled (src/primitives.h:104-114)
  Point(const std::string& s, T scale = 1) {
    static const std::regex r("\\[(\\-?[\\d.]+);\\s*(\\-?[\\d.]+);\\s*(\\-?[\\d.]+)\\]");
    std::smatch m;
    std::regex_search(s, m, r);
    if (m.size() != 4) throw std::invalid_argument("Unknown format of point");
    for (int i = 0; i < 3; ++i) c[i] = std::stod(m[i + 1]) * scale;
  }
  T x() const { return c[0]; }
  T y() const { return c[1]; }
  T z() const { return c[2]; }
  void set(T x, T y, T z) { c[0] = x; c[1] = y; c[2] = z; }
  void setPosition(unsigned pos, T v) {
    if (pos < 3) c[pos] = v; else if (pos == 3) Yaw = v; else if (pos == 4) Pitch = v; else if (pos == 5) Roll = v;
  }
  const T* operator()() const { return c; }
  T operator[](int i) const { return i < 3 ? c[i] : (i == 3 ? Yaw : (i == 4 ? Pitch : (i == 5 ? Roll : T(1)))); }
  friend bool operator==(const Point& a, const Point& b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.Yaw == b.Yaw && a.Pitch == b.Pitch && a.Roll == b.Roll;
  }
  friend bool operator!=(const Point& a, const Point& b) { return !(a == b); }
  // scales the position, NOT the rotation (src/primitives.h:215-222)
  friend Point operator/(const Point& p, const T scale) {
    Point q{p};
    for (int i = 0; i < 3; ++i) q.c[i] /= scale;
    return q;
  }
  void toArray(double out[6]) const { for (int i = 0; i < 6; ++i) out[i] = (*this)[i]; }
  void printPosOnly(std::ostream& out) const { out << c[0] << DELIMITER_OUT << c[1] << DELIMITER_OUT << c[2]; }

 private:
  T c[3];
};

template <class T>
std::ostream& operator<<(std::ostream& out, const Point<T>& p) {
  return out << p.x() << DELIMITER_OUT << p.y() << DELIMITER_OUT << p.z() << DELIMITER_OUT << p.Yaw << DELIMITER_OUT
             << p.Pitch << DELIMITER_OUT << p.Roll;
}

// Only the two process-wide knobs the XML parser writes (src/primitives.h:443-445) plus the fields
// the writers read; the growing forest itself lives inside libsffgpu.
template <class T, class R = Point<T>>
class Node {
 public:
  inline static char ThresholdMisses = DEFAULT_THRES_MISS;
  inline static double SamplingDistance = DEFAULT_SAMP_DIST;
  R Position;
  int Id{0}, ParentId{-1}, TreeId{0};
  unsigned Age{0};
  T DistanceToClosest{0}, DistanceToRoot{0};
  int GetId() const { return Id; }
  unsigned GetAge() const { return Age; }
  bool IsRoot() const { return DistanceToRoot == 0; }
};

// src/primitives.h:680-697
inline int parseString(std::string& inp, std::string& outp1, std::string& outp2, std::string& delimiter) {
  size_t pos = inp.find(delimiter);
  if (pos != std::string::npos) {
    outp1 = inp.substr(0, pos);
    outp2 = inp.substr(pos + delimiter.size());
    return (int)pos;
  }
  outp1 = inp;
  outp2 = "";
  return -1;
}

// src/primitives.h:699-710
inline FileStruct prefixFileName(const FileStruct& path, const std::string& insert) {
  FileStruct r{path};
  auto pos = r.fileName.find_last_of("//");
  if (pos != std::string::npos) r.fileName.insert(pos + 1, insert); else r.fileName.insert(0, insert);
  return r;
}

inline std::string trim(const std::string& s) {
  const char* ws = " \n\r\t\f\v";
  size_t b = s.find_first_not_of(ws);
  if (b == std::string::npos) return "";
  return s.substr(b, s.find_last_not_of(ws) - b + 1);
}
