// match_order.h -- the two orders of match records that the extension stage's culling and the join of reference blocks share
// (/root/reference/src/align/extend.h:51-56 Match::cmp_evalue / cmp_score; output/join_blocks.cpp:129-142 JoinRecord)
#pragma once
#include "../../include/diamond_hip.h"

namespace dmnd {

inline bool match_less(const dmnd_match& a, const dmnd_match& b)    // Match::cmp_evalue, extend.h:51-56
{
	return a.evalue < b.evalue || (a.evalue == b.evalue && (a.hsp.score > b.hsp.score || (a.hsp.score == b.hsp.score && a.target < b.target)));
}

inline bool match_less_score(const dmnd_match& a, const dmnd_match& b)      // Match::cmp_score
{
	return a.hsp.score > b.hsp.score || (a.hsp.score == b.hsp.score && a.target < b.target);
}

}  // namespace dmnd
