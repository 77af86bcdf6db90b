// Small dense linear algebra for the host-side model analysis (N <= a few dozen).
#pragma once
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

namespace bild {
namespace la {

using Mat = std::vector<double>; // row-major

inline double max_abs(const Mat &A)
{
    double m = 0.0;
    for (double v : A) m = std::max(m, std::fabs(v));
    return m;
}

inline double fro(const Mat &A)
{
    double s = 0.0;
    for (double v : A) s += v * v;
    return std::sqrt(s);
}

// C (m x n) = A (m x k) * B (k x n)
inline Mat matmul(const Mat &A, const Mat &B, int m, int k, int n)
{
    Mat C((size_t)m * n, 0.0);
    for (int i = 0; i < m; ++i)
        for (int l = 0; l < k; ++l) {
            const double a = A[(size_t)i * k + l];
            if (a == 0.0) continue;
            for (int j = 0; j < n; ++j) C[(size_t)i * n + j] += a * B[(size_t)l * n + j];
        }
    return C;
}

inline Mat transpose(const Mat &A, int m, int n)
{
    Mat T((size_t)m * n);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) T[(size_t)j * m + i] = A[(size_t)i * n + j];
    return T;
}

// Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix.
// On return: ev[i] eigenvalues, V (n x n) with eigenvectors in COLUMNS; sorted descending.
inline void jacobi_eigh(Mat A, int n, std::vector<double> &ev, Mat &V)
{
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    auto a = [&](int i, int j) -> double & { return A[(size_t)i * n + j]; };
    // symmetrise
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) a(i, j) = a(j, i) = 0.5 * (a(i, j) + a(j, i));
    const double scale = std::max(fro(A), 1e-300);
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) off += a(i, j) * a(i, j);
        if (std::sqrt(off) <= 1e-18 * scale) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = a(p, q);
                if (apq == 0.0) continue;
                const double app = a(p, p), aqq = a(q, q);
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = a(k, p), akq = a(k, q);
                    a(k, p) = c * akp - s * akq;
                    a(k, q) = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = a(p, k), aqk = a(q, k);
                    a(p, k) = c * apk - s * aqk;
                    a(q, k) = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int x, int y) { return a(x, x) > a(y, y); });
    ev.resize(n);
    Mat Vs((size_t)n * n);
    for (int j = 0; j < n; ++j) {
        ev[j] = a(order[j], order[j]);
        for (int k = 0; k < n; ++k) Vs[(size_t)k * n + j] = V[(size_t)k * n + order[j]];
    }
    V.swap(Vs);
}

// Orthonormal basis (columns) of span(cols of X (N x m)) by modified Gram-Schmidt with
// re-orthogonalisation; columns whose remainder is below tol * (their original norm) are
// dropped.  Appends to an existing orthonormal basis Vb (N x nb).
inline void extend_basis(Mat &Vb, int &nb, int N, const Mat &X, int m, double tol)
{
    for (int j = 0; j < m; ++j) {
        std::vector<double> v(N);
        double norm0 = 0.0;
        for (int i = 0; i < N; ++i) {
            v[i] = X[(size_t)i * m + j];
            norm0 += v[i] * v[i];
        }
        norm0 = std::sqrt(norm0);
        if (norm0 == 0.0) continue;
        for (int pass = 0; pass < 2; ++pass)
            for (int b = 0; b < nb; ++b) {
                double dot = 0.0;
                for (int i = 0; i < N; ++i) dot += Vb[(size_t)b * N + i] * v[i];
                for (int i = 0; i < N; ++i) v[i] -= dot * Vb[(size_t)b * N + i];
            }
        double norm = 0.0;
        for (int i = 0; i < N; ++i) norm += v[i] * v[i];
        norm = std::sqrt(norm);
        if (norm <= tol * norm0) continue;
        // basis vectors are stored as ROWS of Vb here (nb x N) for cache friendliness
        Vb.resize((size_t)(nb + 1) * N);
        for (int i = 0; i < N; ++i) Vb[(size_t)nb * N + i] = v[i] / norm;
        ++nb;
    }
}

} // namespace la
} // namespace bild
