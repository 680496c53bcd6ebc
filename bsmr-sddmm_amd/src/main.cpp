// Command-line driver with the call order of the reference's src/main.cu:6-42:
//   BSMR-sddmm -f <matrix.mtx|.smtx|.txt> -k <K> [-a alpha] [-d delta] [-t 1 -l logdir/]
// Extra (MI355X): environment BSMR_COMPUTE = f16 | bf16 | f32, BSMR_VALIDATE=1.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "Logger.hpp"
#include "Matrix.hpp"
#include "Options.hpp"
#include "bsmr_hip.h"
#include "sddmm.hpp"
#include "sddmmKernel.hpp"

int main(int argc, char* argv[]) {
    Options options(argc, argv);

    if (const char* mode = std::getenv("BSMR_COMPUTE")) {
        if (!strcmp(mode, "bf16")) setSddmmComputeMode(BSMR_COMPUTE_BF16);
        else if (!strcmp(mode, "f32")) setSddmmComputeMode(BSMR_COMPUTE_F32);
        else setSddmmComputeMode(BSMR_COMPUTE_F16);
    }

    sparseMatrix::CSR<float> matrixS;
    if (!matrixS.initializeFromMatrixFile(options.inputFile())) {
        fprintf(stderr, "Error, matrix S initialize failed.\n");
        return -1;
    }

    if (options.testMode()) {
        sddmm_testMode(options, matrixS);
        return 0;
    }

    const size_t K = options.K();
    Matrix<float> matrixA(matrixS.row(), static_cast<UIN>(K), row_major);
    matrixA.makeData();
    Matrix<float> matrixB(static_cast<UIN>(K), matrixS.col(), col_major);
    matrixB.makeData();

    Logger logger;
    logger.getInformation(options);
    logger.getInformation(matrixS);
    logger.getInformation(matrixA, matrixB);

    sparseMatrix::CSR<float> matrixP(matrixS);
    sddmm(options, matrixA, matrixB, matrixP, logger);

    logger.printLogInformation();
    // a failed device call (no plan, unsupported K, out of memory, HIP error) is not a result: the record above
    // carries [mi355x_status : n] and the exit code is non-zero
    return logger.status_ == 0 ? 0 : 3;
}
