/* fastx_dump — host-only test helper: prints what the buildG input stage keeps.
 *   fastx_dump <min_overlap> [-pe a,b] [-se c]   ->   one line per good read: <file index>\t<sequence>
 *   last line: "#records <total> stride <words>"                                                                  */
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "fastx.h"

int main(int argc, char **argv)
{
    if (argc < 2) return 1;
    uint32_t mo = (uint32_t)atoi(argv[1]);
    std::vector<std::string> pe, se;
    for (int i = 2; i + 1 < argc; i += 2) {
        std::stringstream ss(argv[i + 1]);
        std::string item;
        while (std::getline(ss, item, ',')) (std::string(argv[i]) == "-pe" ? pe : se).push_back(item);
    }
    disco::ReadSet rs;
    std::string err;
    if (!disco::load_reads(pe, se, mo, 4, rs, err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return 2;
    }
    const std::vector<uint64_t> woff = rs.word_offsets();
    for (uint64_t i = 0; i < rs.size(); i++) {
        std::string s(rs.len[i], 'A');
        for (uint32_t t = 0; t < rs.len[i]; t++) s[t] = "ACGT"[(rs.packed[woff[i] + (t >> 5)] >> (62 - 2 * (t & 31))) & 3];
        printf("%llu\t%s\n", (unsigned long long)rs.file_index[i], s.c_str());
    }
    printf("#records %llu stride %u\n", (unsigned long long)rs.total_records, rs.stride_words);
    return 0;
}
