// sff_dropin.h — everything the reference's unmodified src/main.cpp expects from its "main.h"
// (reference src/main.h:14-35): the standard headers it uses unqualified, the vendored rapidxml parser
// (third party, taken from the reference tree's lib/rapidxml at build time), the solver classes of this
// header set - which forward Solve() to libsffgpu - the two using-directives its code relies on, and the
// prototypes of the two functions main.cpp itself defines.
#pragma once

// --- solver side: Problem<T>, Environment<T>, SpaceForest / RapidExpTree / LazyTSP on top of the C ABI
#include "primitives.h"
#include "environment.h"
#include "problemStruct.h"
#include "forest.h"
#include "rrt.h"
#include "lazy.h"

// --- parser side (main.cpp is the XML front end; it is compiled as is)
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>

#include "rapidxml.hpp"

using namespace rapidxml;   // main.cpp names xml_document / xml_node / xml_attribute unqualified
using namespace std;        // ... and string, cout, stoi, make_unique

bool getFile(rapidxml::xml_node<>* node, FileStruct& file, int iteration = 0, bool includeIter = true);
void parseFile(const std::string& fileName, Problem<double>& problem);
