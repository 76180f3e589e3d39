// oracle/ref_graph_dump.cpp -- TEST INFRASTRUCTURE ONLY, container-only.
// Driver around the REAL reference graph model: it is compiled together with
// /root/reference/src/{Graph,Segment,Vertex,Edge,Junction,Weight,Exceptions}.cpp (they build from their own
// sources with g++, no external libraries) into oracle/_ref/ref_graph_dump.  It calls only the reference's public
// API -- Graph(const char*) (Graph.cpp:36-49), calculateHapDepth (:312), calculateCopyNum (:369), getSegments,
// getJunctions -- and prints the parsed graph as JSON in the shape of oracle_graph_dump() so the restated reader
// (#1-#3 of SURVEY 8a) is pinned against the reference itself.  No reference source is copied into this repo.
// With a second argument (a .juncs file) it also drives the GRAPH-LEVEL effects of LocalGenomicMap::readComponents
// (LGM.cpp:5096-5156) through the reference's own graph API: for every strand / partition break of a line it builds the
// probe junction, asks Graph::findJunction (Graph.cpp:501-511) and either Graph::addJunction (coverage =
// Graph::getAvgCoverage(), copy number 1) or raises the found junction's copy number to 2 -- the calls LGM.cpp:5133-5141
// makes, in that order.  The loop around those calls is this driver's (LocalGenomicMap.cpp itself cannot be compiled
// here), so what this pins is the reference's matching rule, duplicate handling, coverage value and junction order.
// With REF_WRITE_LH=<path> in the environment it also calls Graph::writeGraph (Graph.cpp:239-266) at the end: the fixtures the
// product's .lh writer is pinned with (tests/golden/written_*.lh).
// LocalGenomicMap.cpp (the BFB stages) is NOT buildable here: it includes <coin/CbcModel.hpp> and
// <coin/OsiClpSolverInterface.hpp>, which the image lacks.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <vector>

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
    if (argc > 2) {
        // partitions as localhap.cpp:90-98 sets them: segment ids source..sink of chromosome n
        std::vector<Segment*>& src = *g->getMSources();
        std::vector<Segment*>& snk = *g->getMSinks();
        for (size_t n = 0; n < src.size(); n++)
            for (int id = src[n]->getId(); id <= snk[n]->getId(); id++) g->getSegmentById(id)->setPartition((int)n);
        std::ifstream in(argv[2]);
        std::string text;
        while (std::getline(in, text)) {
            std::istringstream ls(text);
            std::vector<int> ids; std::vector<char> strands;
            std::string tok;
            while (ls >> tok) { ids.push_back(std::stoi(tok.substr(0, tok.size() - 1))); strands.push_back(tok[tok.size() - 1]); }
            for (size_t k = 1; k < ids.size(); k++) {
                const bool brk = g->getSegmentById(ids[k - 1])->getPartition() != g->getSegmentById(ids[k])->getPartition() ||
                                 strands[k - 1] != strands[k];
                // (LGM.cpp:5118 compares with the segment at the last break; every token since that break lies on its
                // chromosome -- a change would have been a break -- so comparing with the previous token is the same test)
                if (!brk) continue;
                Junction* probe = new Junction(g->getSegmentById(ids[k - 1]), g->getSegmentById(ids[k]), strands[k - 1], strands[k],
                                               g->getAvgCoverage(), 1, 1, false, true, false);
                Junction* hit = g->findJunction(probe);
                if (hit == NULL) g->addJunction(ids[k - 1], strands[k - 1], ids[k], strands[k], g->getAvgCoverage(), 1, 1, false, true, false);
                else if (hit->getWeight()->getCopyNum() < 2) hit->getWeight()->setCopyNum(2);
            }
        }
    }
    // REF_WRITE_LH=<path>: also run the reference's Graph::writeGraph (Graph.cpp:239-266) on the graph as it stands now
    if (const char* w = getenv("REF_WRITE_LH")) g->writeGraph(w);
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
