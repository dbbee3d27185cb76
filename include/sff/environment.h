// environment.h — source-compatible Environment / Obstacle (reference src/environment.h) on top of
// libsffgpu: the meshes are parsed with the reference's quirks (:125-223), kept as flat triangle
// arrays, and Environment::Collide (:306-316) is answered by the GPU collision kernel.
#pragma once
#include <cfloat>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "sff_gpu.h"
#include "primitives.h"

template <class T> class Obstacle;


template <class T>
class Environment {
 public:
  std::deque<Obstacle<T>> Obstacles;
  Obstacle<T>* Robot;
  Range<T> limits{DBL_MAX, -DBL_MAX, DBL_MAX, -DBL_MAX, DBL_MAX, -DBL_MAX};
  bool HasMap{true};
  T ScaleFactor{1};

  Environment() : Robot{nullptr} {}
  ~Environment() { delete Robot; }

  void processLimits(Range<T>& l) {   // src/environment.h:46-53
    limits.minX = std::min(limits.minX, l.minX); limits.maxX = std::max(limits.maxX, l.maxX);
    limits.minY = std::min(limits.minY, l.minY); limits.maxY = std::max(limits.maxY, l.maxY);
    limits.minZ = std::min(limits.minZ, l.minZ); limits.maxZ = std::max(limits.maxZ, l.maxZ);
  }

  // uploads the robot and the merged obstacles once (replaces the RAPID models)
  void upload() {
    if (uploaded) return;
    std::vector<double> env;
    if (HasMap)
      for (Obstacle<T>& o : Obstacles) env.insert(env.end(), o.triangles().begin(), o.triangles().end());
    sff_compat::check(sffgpu_mesh_upload(sff_compat::gpu(), SFFGPU_MESH_ENV, env.data(), (int)(env.size() / 9)), "env mesh");
    if (!Robot) { std::cout << "Environment: no robot model\n"; std::exit(1); }
    sff_compat::check(sffgpu_mesh_upload(sff_compat::gpu(), SFFGPU_MESH_ROBOT, Robot->triangles().data(),
                                         (int)(Robot->triangles().size() / 9)), "robot mesh");
    uploaded = true;
  }

  bool Collide(Point<T> position) {   // src/environment.h:306-316
    if (!HasMap) return false;
    upload();
    double p[6];
    position.toArray(p);
    uint8_t hit = 0;
    sff_compat::check(sffgpu_collide_poses(sff_compat::gpu(), p, 1, &hit), "collide");
    return hit != 0;
  }

 private:
  bool uploaded{false};
};

template <class T>
class Obstacle {
 public:
  inline static std::string Delimiter = " ";
  inline static std::string NameDelimiter = "_";
  Point<T> Position;

  Obstacle() {}
  Obstacle(const std::string fileName, const bool isObj, const T scaleFactor) : Obstacle(fileName, isObj, Point<T>(), scaleFactor) {}
  Obstacle(const std::string fileName, const bool isObj, const Point<T> position, const T scaleFactor)
      : Position{position}, scale{scaleFactor} {
    if (isObj) ParseOBJFile(fileName); else ParseMapFile(fileName);
  }
  virtual ~Obstacle() {}

  Range<T>& getRange() { return localRange; }
  const std::vector<double>& triangles() const { return tris; }

  // src/environment.h:125-166: any line whose first token starts with 'v' is a vertex (so "vn" lines
  // are vertices too), faces are parsed with stoi ("1//1" -> 1), group offsets never advance
  void ParseOBJFile(const std::string fileName) {
    std::ifstream f(fileName);
    std::string line, value;
    while (getline(f, line)) {
      parseString(line, value, line, Delimiter);
      switch (value.empty() ? '\0' : value[0]) {
        case 'v': {
          T p[3];
          for (int i = 0; i < 3; ++i) {
            parseString(line, value, line, Delimiter);
            p[i] = std::stod(value) + Position[i];
          }
          addPoint(p);
          break;
        }
        case 'f': {
          int k[3];
          for (int i = 0; i < 3; ++i) {
            parseString(line, value, line, Delimiter);
            k[i] = std::stoi(value);
          }
          addFacet(k);
          break;
        }
        default: break;
      }
    }
  }
  // src/environment.h:169-195: rows of 3 x (x y), z = 0
  void ParseMapFile(const std::string fileName) {
    std::ifstream f(fileName);
    std::string line, value;
    int index = 1;
    while (getline(f, line)) {
      line = trim(line);
      if (line.empty()) continue;
      int k[3];
      for (int i = 0; i < 3; ++i) {
        T p[3] = {0, 0, 0};
        for (int j = 0; j < 2; ++j) {
          parseString(line, value, line, Delimiter);
          p[j] = std::stod(value) + Position[j];
        }
        addPoint(p);
        k[i] = index + i;
      }
      index += 3;
      addFacet(k);
    }
  }

 protected:
  std::vector<T> pts;          // facePoints, flat xyz
  std::vector<double> tris;    // faces, 9 doubles each
  Range<T> localRange{DBL_MAX, -DBL_MAX, DBL_MAX, -DBL_MAX, DBL_MAX, -DBL_MAX};
  T scale{1};

  void addPoint(T c[3]) {       // src/environment.h:197-210: scaled AFTER the position was added
    for (int i = 0; i < 3; ++i) { c[i] *= scale; pts.push_back(c[i]); }
    localRange.minX = std::min(localRange.minX, c[0]); localRange.maxX = std::max(localRange.maxX, c[0]);
    localRange.minY = std::min(localRange.minY, c[1]); localRange.maxY = std::max(localRange.maxY, c[1]);
    localRange.minZ = std::min(localRange.minZ, c[2]); localRange.maxZ = std::max(localRange.maxZ, c[2]);
  }
  void addFacet(int k[3]) {     // src/environment.h:212-223
    for (int i = 0; i < 3; ++i) {
      size_t at = (size_t)(k[i] - 1) * 3;
      if (k[i] < 1 || at + 2 >= pts.size()) { std::cout << "Obstacle: face index out of range\n"; std::exit(1); }
      for (int j = 0; j < 3; ++j) tris.push_back((double)pts[at + j]);
    }
  }
};
