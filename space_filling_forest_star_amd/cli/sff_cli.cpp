// sff_cli.cpp — stand-alone command-line front end of libsffgpu: reads the planner's XML configuration
// (schema: reference README.md:42-272, semantics: reference src/main.cpp:40-462) and runs the solver of the
// drop-in header set (include/sff/), so that existing configs run without the reference tree:
//
//     sff_cli config.xml [iteration]
//
// Same contract as the reference binary (src/main.cpp:14-38): exit status 2 without arguments, messages on
// stdout and exit(1) on a missing file or a rejected configuration, the optional second argument is the run
// number that gets spliced into the output file names (getFile, src/main.cpp:464-495).  The XML reader below
// is a small subset parser written for this schema (elements, attributes, comments, declarations); it replaces
// the reference's vendored rapidxml.  Environment switches: SFF_SEED, SFF_WAVE, SFFGPU_DEVICE (include/sff/).
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "primitives.h"
#include "environment.h"
#include "problemStruct.h"
#include "forest.h"
#include "rrt.h"
#include "lazy.h"

namespace {

// ---------------------------------------------------------------- XML subset
struct XmlNode {
  std::string name;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<std::unique_ptr<XmlNode>> kids;
  XmlNode* parent = nullptr;
  int index_in_parent = 0;

  const std::string* attr(const char* key) const {
    for (const auto& a : attrs)
      if (a.first == key) return &a.second;
    return nullptr;
  }
  const XmlNode* child(const char* key) const {
    for (const auto& k : kids)
      if (k->name == key) return k.get();
    return nullptr;
  }
  // the element after this one under the same parent, whatever its name (the reference walks obstacles and
  // points with rapidxml's unnamed next_sibling(), src/main.cpp:264,290)
  const XmlNode* next() const {
    if (!parent || index_in_parent + 1 >= (int)parent->kids.size()) return nullptr;
    return parent->kids[index_in_parent + 1].get();
  }
};

class XmlReader {
 public:
  explicit XmlReader(const std::string& text) : s(text) {}
  std::unique_ptr<XmlNode> parse() {
    auto doc = std::make_unique<XmlNode>();
    while (true) {
      skip_misc();
      if (at_end()) break;
      if (s[i] != '<') fail("text outside of an element");
      add_child(doc.get(), element());
    }
    return doc;
  }

 private:
  const std::string& s;
  size_t i = 0;
  bool at_end() const { return i >= s.size(); }
  [[noreturn]] void fail(const char* what) const { throw std::invalid_argument(std::string("malformed XML: ") + what); }
  static void add_child(XmlNode* p, std::unique_ptr<XmlNode> c) {
    c->parent = p;
    c->index_in_parent = (int)p->kids.size();
    p->kids.push_back(std::move(c));
  }
  bool starts(const char* lit) const { return s.compare(i, std::strlen(lit), lit) == 0; }
  void skip_until(const char* lit) {
    size_t e = s.find(lit, i);
    if (e == std::string::npos) fail("unterminated construct");
    i = e + std::strlen(lit);
  }
  // white space, text, comments, declarations and processing instructions between elements
  void skip_misc() {
    while (!at_end()) {
      if (starts("<!--")) skip_until("-->");
      else if (starts("<?")) skip_until("?>");
      else if (starts("<!")) skip_until(">");
      else if (s[i] == '<') return;
      else ++i;   // character data carries nothing in this schema
    }
  }
  std::string name() {
    size_t b = i;
    while (!at_end() && (std::isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == '-' || s[i] == ':' || s[i] == '.')) ++i;
    if (b == i) fail("name expected");
    return s.substr(b, i - b);
  }
  void spaces() { while (!at_end() && std::isspace((unsigned char)s[i])) ++i; }
  static std::string unescape(const std::string& v) {
    static const std::pair<const char*, char> ent[] = {{"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
    std::string o;
    for (size_t k = 0; k < v.size();) {
      bool hit = false;
      if (v[k] == '&')
        for (const auto& e : ent) {
          size_t n = std::strlen(e.first);
          if (v.compare(k, n, e.first) == 0) { o += e.second; k += n; hit = true; break; }
        }
      if (!hit) o += v[k++];
    }
    return o;
  }
  std::unique_ptr<XmlNode> element() {
    ++i;   // '<'
    auto n = std::make_unique<XmlNode>();
    n->name = name();
    while (true) {
      spaces();
      if (at_end()) fail("unterminated start tag");
      if (starts("/>")) { i += 2; return n; }
      if (s[i] == '>') { ++i; break; }
      std::string key = name();
      spaces();
      if (at_end() || s[i] != '=') fail("'=' expected after an attribute name");
      ++i;
      spaces();
      if (at_end() || (s[i] != '"' && s[i] != '\'')) fail("quoted attribute value expected");
      const char q = s[i++];
      size_t e = s.find(q, i);
      if (e == std::string::npos) fail("unterminated attribute value");
      n->attrs.emplace_back(std::move(key), unescape(s.substr(i, e - i)));
      i = e + 1;
    }
    while (true) {   // content
      skip_misc();
      if (at_end()) fail("missing end tag");
      if (starts("</")) {
        i += 2;
        if (name() != n->name) fail("mismatched end tag");
        spaces();
        if (at_end() || s[i] != '>') fail("'>' expected");
        ++i;
        return n;
      }
      add_child(n.get(), element());
    }
  }
};

// ---------------------------------------------------------------- configuration -> Problem<double>
[[noreturn]] void reject(const std::string& why) { throw std::invalid_argument(why); }

const std::string& need(const XmlNode* n, const char* key, const char* complaint) {
  const std::string* v = n ? n->attr(key) : nullptr;
  if (!v) reject(complaint);
  return *v;
}

// file="..." is_obj="..." of an input or output element; false when the element or its file attribute is
// missing.  Output names get "_<run>" in front of the extension when a run number was given.
bool file_of(const XmlNode* n, FileStruct& out, int run = 0, bool splice_run = true) {
  const std::string* f = n ? n->attr("file") : nullptr;
  if (!f) return false;
  out.fileName = *f;
  if (run != 0 && splice_run) out.fileName.insert(out.fileName.find_last_of('.'), '_' + std::to_string(run));
  const std::string* o = n->attr("is_obj");
  if (!o || *o == "false") out.type = Map;
  else if (*o == "true") out.type = Obj;
  else reject("invalid attribute isObj in file node!");
  return true;
}

void load_range_axis(const XmlNode* range, const char* axis, const char* low_name, double scale, double& lo, double& hi) {
  const XmlNode* n = range->child(axis);
  if (!n) reject(std::string("invalid ") + low_name + " node in range node");
  lo = scale * std::stod(need(n, "min", (std::string("invalid min attribute in ") + low_name + " node").c_str()));
  hi = scale * std::stod(need(n, "max", (std::string("invalid max attribute in ") + low_name + " node").c_str()));
}

void load_outputs(const XmlNode* save, Problem<double>& P) {
  struct Out { const char* element; SaveOptions flag; bool splice_run; };
  static const Out outs[] = {{"Goals", SaveGoals, true},     {"Tree", SaveTree, true},   {"RawPath", SaveRaw, true},
                             {"SmoothPath", SaveSmooth, true}, {"Params", SaveParams, false}, {"TSP", SaveTSP, true},
                             {"Frontiers", SaveFrontiers, true}};
  for (const Out& o : outs) {
    const XmlNode* n = save->child(o.element);
    FileStruct f;
    if (!file_of(n, f, P.iteration, o.splice_run)) continue;
    // (the reference refuses a SmoothPath output exactly when smoothing is ON - an inverted test, src/main.cpp:386-389,
    //  SURVEY.md Appendix A.15; kept, since configs in the field are written against it)
    if (o.flag == SaveSmooth && P.smoothing)
      reject("smoothing is disabled, therefore \"SmoothPath\" parameter might not be defined!");
    if (o.flag == SaveFrontiers && P.solver != SFF) reject("frontiers output is defined only for SFF-based solvers!");
    P.saveOptions = P.saveOptions | o.flag;
    P.fileNames[o.flag] = f;
    if (o.flag == SaveParams)
      if (const std::string* id = n->attr("id")) P.id = *id;
    if (o.flag == SaveTree || o.flag == SaveFrontiers) {
      const std::string* every = n->attr("everyIteration");
      if (every && std::stoi(*every) != 0) {
        P.saveOptions = P.saveOptions | SaveConcurrent;
        (o.flag == SaveTree ? P.saveTreeIter : P.saveFrontiersIter) = std::stoi(*every);
      }
    }
  }
}

void load_problem(const std::string& path, Problem<double>& P) {
  std::ifstream in(path.c_str());
  if (!in.good() || !in.is_open()) {
    std::cout << "Cannot open config file at: " << path << "\n";
    std::exit(1);
  }
  std::stringstream text;
  text << in.rdbuf();
  const std::string xml = text.str();
  try {
    const std::unique_ptr<XmlNode> doc = XmlReader(xml).parse();
    const XmlNode* root = doc->child("Problem");
    if (!root) reject("invalid root node");

    const std::string& solver = need(root, "solver", "invalid solver attibute in Problem node!");
    if (solver == "sff") P.solver = SFF;
    else if (solver == "rrt") P.solver = RRT;
    else if (solver == "lazy") P.solver = Lazy;
    else reject("unknown solver type in Problem node, use either sff or rrt");
    P.optimal = need(root, "optimize", "invalid optimize attibute in Problem node!") == "true";
    const std::string* a = root->attr("smoothing");
    P.smoothing = a && *a == "true";
    if (P.solver == Lazy && P.smoothing)
      reject("Lazy-RRT* solver with path smoothing is not implemented, set \"smoothing\" parameter to \"false\"");
    a = root->attr("scale");
    const double scale = a ? std::stod(*a) : 1.0;
    a = root->attr("dim");
    if (!a || *a == "3D" || *a == "3d") P.dimension = D3;
    else if (*a == "2D" || *a == "2d") P.dimension = D2;
    else reject("invalid dim attribute!");

    if (const XmlNode* n = root->child("ObjectDelimiters")) {
      if (const std::string* v = n->attr("standard")) Obstacle<double>::Delimiter = *v;
      if (const std::string* v = n->attr("name")) Obstacle<double>::NameDelimiter = *v;
    }

    const XmlNode* tsp = root->child("TSP");
    if (!tsp && P.solver == Lazy) reject("missing TSP solver parameters for Lazy solver!");
    if (tsp) {
      if (P.solver != Lazy)
        std::cout << "Warning: TSP solver is called only in Lazy solver algorithm -- defined TSP parameters are redundant and will not be used.\n";
      P.tspSolver = need(tsp, "path", "invalid path attribute in TSP node!");
      P.tspType = need(tsp, "type", "invalid type attribute in TSP node!");
    }

    const XmlNode* robot = root->child("Robot");
    if (!robot) reject("invalid Robot node!");
    FileStruct rf;
    if (!file_of(robot, rf)) reject("invalid file node in Robot node!");
    P.environment.Robot = new Obstacle<double>(rf.fileName, rf.type == Obj, scale);

    const XmlNode* range = root->child("Range");
    if (!range) reject("invalid range node");
    a = range->attr("autoDetect");
    P.autoRange = a && *a == "true";
    if (!P.autoRange) {   // (with autoDetect the Range* elements are not read at all)
      Range<double>& l = P.environment.limits;
      load_range_axis(range, "RangeX", "rangex", scale, l.minX, l.maxX);
      load_range_axis(range, "RangeY", "rangey", scale, l.minY, l.maxY);
      load_range_axis(range, "RangeZ", "rangez", scale, l.minZ, l.maxZ);
    }

    const XmlNode* envn = root->child("Environment");
    if (!envn) {
      P.environment.HasMap = false;
    } else {
      P.collisionDist = scale * std::stod(need(envn, "collision", "invalid collision attribute in Environment node!"));
      P.environment.ScaleFactor = scale;
      const XmlNode* ob = envn->child("Obstacle");
      if (!ob) P.environment.HasMap = false;
      for (; ob; ob = ob->next()) {
        FileStruct f;
        if (!file_of(ob, f)) reject("invalid file attribute in Obstacle node!");
        const std::string* where = ob->attr("position");
        const Point<double> at = where ? Point<double>(*where) : Point<double>();
        Obstacle<double>& o = P.environment.Obstacles.emplace_back(f.fileName, f.type == Obj, at, scale);
        if (P.autoRange) P.environment.processLimits(o.getRange());
      }
    }

    const XmlNode* points = root->child("Points");
    if (!points) reject("invalid Points node - insert at least one point!");
    const XmlNode* pt = points->child("Point");
    if (!pt) reject("invalid Point subnode in Points node - insert at least one point!");
    for (; pt; pt = pt->next()) P.roots.emplace_back(need(pt, "coord", "invalid coord attribute in Point node!"), scale);
    if (P.solver == RRT && P.optimal && P.roots.size() > 1) reject("Multi-T-RRT* is undefined!");

    if (const XmlNode* goal = root->child("Goal")) {
      if (P.solver == Lazy) reject("single point path planning not defined for Lazy solver (use RRT/RRT* solver instead)!");
      if (P.roots.size() > 1) std::cout << "Warning: Multi-source planning with one goal has not been tested!\n";
      P.hasGoal = true;
      P.goal = Point<double>(need(goal, "coord", "invalid coord attribute in Goal node!"), scale);
    }

    const XmlNode* dist = root->child("Distances");
    if (!dist) reject("invalid Distances node!");
    P.distTree = scale * std::stod(need(dist, "dtree", "invalid dtree attribute in Distances node!"));
    Node<double, Point<double>>::SamplingDistance = scale * std::stod(need(dist, "circum", "invalid circum attribute in Distances node!"));

    if (const XmlNode* imp = root->child("Improvements")) {
      if (const std::string* v = imp->attr("priorityBias")) P.priorityBias = std::stod(*v);
      if (!P.hasGoal && P.priorityBias != 0 && P.solver == RRT) reject("Multi-T-RRT with bias is undefined!");
      if (P.solver == Lazy && P.priorityBias != 0) reject("priority bias for Lazy solver is not implemented!");
    }
    if (const XmlNode* th = root->child("Thresholds"))
      if (const std::string* v = th->attr("standard")) Node<double>::ThresholdMisses = std::stoi(*v);

    const XmlNode* it = root->child("MaxIterations");
    if (!it) reject("invalid MaxIterations node");
    P.maxIterations = std::stoi(need(it, "value", "invalid value attribute in MaxIterations node"));

    if (const XmlNode* save = root->child("Save")) load_outputs(save, P);
  } catch (const std::invalid_argument& e) {
    std::cout << "Problem loading error: " << e.what() << "\n";
    std::exit(1);
  }
}

}  // namespace

int main(int argc, char* argv[]) {
  if (argc < 2) return 2;
  Problem<double> problem;
  if (argc == 3) problem.iteration = std::stoi(argv[2]);
  load_problem(argv[1], problem);
  std::unique_ptr<Solver<double, Point<double>>> solver;
  switch (problem.solver) {
    case SFF: solver = std::make_unique<SpaceForest<double, Point<double>>>(problem); break;
    case RRT: solver = std::make_unique<RapidExpTree<double, Point<double>>>(problem); break;
    case Lazy: solver = std::make_unique<LazyTSP<double, Point<double>>>(problem); break;
  }
  solver->Solve();
  return 0;
}
