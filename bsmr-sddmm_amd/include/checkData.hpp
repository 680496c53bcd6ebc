#pragma once
// Element-wise acceptance test of the reference (include/checkData.hpp:14,21-79):
// pass iff |a-b| < 1e-5 or |a-b| / max(|a|,|b|,1e-3) < 1e-3; integers compare
// exactly.  Prints the same report block and the first nine mismatches.

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <type_traits>
#include <vector>

const float ERROR_THRESHOLD_EPSILON = 1e-3f;

template <typename T>
inline bool checkOneData(const T data1, const T data2) {
    if constexpr (std::is_floating_point<T>::value) {
        const T absDiff = std::fabs(data1 - data2);
        if (absDiff < static_cast<T>(1e-5)) return true;
        const T scale = std::max(std::max(std::fabs(data1), std::fabs(data2)),
                                 static_cast<T>(ERROR_THRESHOLD_EPSILON));
        return absDiff / scale < static_cast<T>(ERROR_THRESHOLD_EPSILON);
    } else {
        return data1 == data2;
    }
}

template <typename T>
inline bool checkDataFunction(const size_t num, const T* data1, const T* data2, size_t& numError) {
    printf("|---------------------------check data---------------------------|\n");
    printf("| Data size : %zu\n", num);
    printf("| Error threshold epsilon : %f\n", ERROR_THRESHOLD_EPSILON);
    printf("| Checking results...\n");
    size_t errors = 0;
    for (size_t i = 0; i < num; ++i) {
        if (checkOneData(data1[i], data2[i])) continue;
        if (++errors < 10)
            printf("| Error : idx = %zu, data1 = %f, data2 = %f, difference = %f\n", i,
                   static_cast<float>(data1[i]), static_cast<float>(data2[i]),
                   static_cast<float>(data1[i] - data2[i]));
    }
    numError = errors;
    if (errors)
        printf("| No Pass! Inconsistent data! %zu errors! Error rate : %2.2f%%\n", errors,
               static_cast<float>(errors) / static_cast<float>(num) * 100);
    else
        printf("| Pass! Result validates successfully.\n");
    printf("|----------------------------------------------------------------|\n");
    return errors == 0;
}

template <typename T>
inline bool checkData(const std::vector<T>& a, const std::vector<T>& b, size_t& numError) {
    if (a.size() != b.size()) return false;
    return checkDataFunction(a.size(), a.data(), b.data(), numError);
}

template <typename T>
inline bool checkData(const std::vector<T>& a, const std::vector<T>& b) {
    size_t numError = 0;
    return checkData(a, b, numError);
}
