// main.h — what the reference's src/main.cpp includes (src/main.h:14-35), resolved to the drop-in
// header set.  rapidxml comes from the reference tree's vendored lib/rapidxml (third party).
#pragma once
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>

#include "rapidxml.hpp"

#include "primitives.h"
#include "environment.h"
#include "problemStruct.h"
#include "forest.h"
#include "rrt.h"
#include "lazy.h"

using namespace std;
using namespace rapidxml;

void parseFile(const std::string& fileName, Problem<double>& problem);
bool getFile(rapidxml::xml_node<>* node, FileStruct& file, int iteration = 0, bool includeIter = true);
