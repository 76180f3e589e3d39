// oracle/ref_graph_dump.cpp -- TEST INFRASTRUCTURE ONLY, container-only.
// Driver around the REAL reference graph model: it is compiled together with
// /root/reference/src/{Graph,Segment,Vertex,Edge,Junction,Weight,Exceptions}.cpp (they build from their own
// sources with g++, no external libraries) into oracle/_ref/ref_graph_dump.  It calls only the reference's public
// API -- Graph(const char*) (Graph.cpp:36-49), calculateHapDepth (:312), calculateCopyNum (:369), getSegments,
// getJunctions -- and prints the parsed graph as JSON in the shape of oracle_graph_dump() so the restated reader
// (#1-#3 of SURVEY 8a) is pinned against the reference itself.  No reference source is copied into this repo.
// LocalGenomicMap.cpp (the BFB stages) is NOT buildable here: it includes <coin/CbcModel.hpp> and
// <coin/OsiClpSolverInterface.hpp>, which the image lacks.
#include <iostream>
#include <sstream>

#include "Graph.hpp"

static void jstr(std::ostream& o, const std::string& s) { o << '"' << s << '"'; }

int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "usage: ref_graph_dump file.lh\n"; return 2; }
    // the reference prints progress to stdout; capture it so the JSON stays clean
    std::ostringstream captured;
    std::streambuf* old = std::cout.rdbuf(captured.rdbuf());
    Graph* g = new Graph(argv[1]);
    g->calculateHapDepth();
    g->calculateCopyNum();
    std::cout.rdbuf(old);

    std::ostringstream o;
    o.precision(17);
    o << "{\"ok\":true,\"err\":\"\",\"segs\":[";
    bool first = true;
    for (Segment* s : *g->getSegments()) {
        if (!first) o << ',';
        first = false;
        o << '[' << s->getId() << ',' << s->getChrId() << ','; jstr(o, s->getChrom());
        o << ',' << s->getStart() << ',' << s->getEnd() << ',' << s->getWeight()->getCoverage() << ','
          << s->getWeight()->getCopyNum() << ']';
    }
    o << "],\"juncs\":[";
    first = true;
    for (Junction* j : *g->getJunctions()) {
        if (!first) o << ',';
        first = false;
        o << '[' << j->getSource()->getId() << ",\"" << j->getSourceDir() << "\"," << j->getTarget()->getId() << ",\""
          << j->getTargetDir() << "\"," << j->getWeight()->getCoverage() << ',' << j->getWeight()->getCopyNum() << ','
          << (j->isInferred() ? 1 : 0) << ',' << (j->hasLowerBoundLimit() ? 1 : 0) << ']';
    }
    o << "],\"sources\":[";
    first = true;
    for (Segment* s : *g->getMSources()) { if (!first) o << ','; first = false; o << s->getId(); }
    o << "],\"sinks\":[";
    first = true;
    for (Segment* s : *g->getMSinks()) { if (!first) o << ','; first = false; o << s->getId(); }
    o << "],\"log\":[";
    first = true;
    std::istringstream lines(captured.str());
    std::string line;
    while (std::getline(lines, line)) { if (!first) o << ','; first = false; jstr(o, line); }
    o << "]}";
    std::cout << o.str() << std::endl;
    return 0;
}
