// Test stand-ins for the two DBoW2 containers the ORB front end touches
// (Thirdparty/DBoW2/DBoW2/BowVector.h, FeatureVector.h): std::map subclasses with addWeight / normalize and
// addFeature.  Used only by the test build of the drop-in classes; a real tree includes DBoW2's own headers.
#ifndef ORBHIP_STANDIN_DBOW2_H
#define ORBHIP_STANDIN_DBOW2_H
#include <map>
#include <vector>

namespace DBoW2 {
enum LNorm { L1, L2 };
struct BowVector : std::map<unsigned int, double> {
    void addWeight(unsigned int id, double v) {
        iterator vit = lower_bound(id);
        if (vit != end() && !(key_comp()(id, vit->first)))
            vit->second += v;
        else
            insert(vit, value_type(id, v));
    }
    void normalize(LNorm) {  // L1
        double norm = 0.0;
        for (iterator it = begin(); it != end(); ++it) norm += it->second < 0 ? -it->second : it->second;
        if (norm > 0.0)
            for (iterator it = begin(); it != end(); ++it) it->second /= norm;
    }
};
struct FeatureVector : std::map<unsigned int, std::vector<unsigned int> > {
    void addFeature(unsigned int id, unsigned int i_feature) {
        iterator vit = lower_bound(id);
        if (vit != end() && vit->first == id)
            vit->second.push_back(i_feature);
        else {
            vit = insert(vit, value_type(id, std::vector<unsigned int>()));
            vit->second.push_back(i_feature);
        }
    }
};
}  // namespace DBoW2
#endif
