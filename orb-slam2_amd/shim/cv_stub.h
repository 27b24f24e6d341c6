// Minimal stand-in for the few OpenCV 2.4 core types the ORBextractor / ORBmatcher interfaces
// mention, so that the drop-in classes can be compiled and tested where OpenCV is not installed.
// With real OpenCV, do not define ORBHIP_SHIM_CV_STUB and this header is not used.
#ifndef ORBHIP_CV_STUB_H
#define ORBHIP_CV_STUB_H
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

namespace cv {

struct Point2f {
    float x, y;
};

struct KeyPoint {  // field-for-field the 2.4 class: 28 bytes
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    unsigned char* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int, void* ext, size_t st) : rows(r), cols(c), step(st), data((unsigned char*)ext) {}
    void create(int r, int c, int) {
        if (r == rows && c == cols && owner_) return;
        owner_.reset(new unsigned char[(size_t)r * c ? (size_t)r * c : 1], std::default_delete<unsigned char[]>());
        data = owner_.get();
        rows = r;
        cols = c;
        step = (size_t)c;
    }
    void release() {
        owner_.reset();
        data = nullptr;
        rows = cols = 0;
        step = 0;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols; }
    int type() const { return CV_8UC1; }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    Mat row(int r) const { return Mat(1, cols, CV_8U, data + (size_t)r * step, step); }

private:
    std::shared_ptr<unsigned char> owner_;
};

class _InputArray {
public:
    _InputArray(const Mat& m) : m_(&m) {}
    Mat getMat() const { return *m_; }
    bool empty() const { return m_->empty(); }

private:
    const Mat* m_;
};
class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int r, int c, int t) const { m_->create(r, c, t); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }

private:
    Mat* m_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;

}  // namespace cv
#endif
