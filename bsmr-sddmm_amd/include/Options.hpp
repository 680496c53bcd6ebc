#pragma once
// Command-line contract of the reference binary (include/Options.hpp:52-69,
// :78-124): -f <file> -k <K> -a <alpha> -d <delta> -t <0|1> -l <logdir>, option
// letters case-insensitive, or the positional form `<file> <K>`.  Own code.

#include <cstddef>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>

#include "util.hpp"

class Options {
public:
    Options(const int argc, const char* const argv[]) {
        if (argc > 0) {
            programPath_ = util::getParentFolderPath(argv[0]);
            programName_ = util::getFileName(argv[0]);
        }
        // Every argument that starts with '-' is an option and consumes the
        // argument after it; a repeated option keeps its first value.
        std::map<std::string, std::string> given;
        for (int i = 1; i < argc; ++i) {
            if (argv[i][0] != '-') continue;
            const std::string opt = argv[i];
            if (given.count(opt)) {
                std::cerr << "Option " << opt << " is duplicated." << std::endl;
                continue;
            }
            if (i + 1 >= argc) {
                std::cerr << "Option " << opt << " requires an argument." << std::endl;
                continue;
            }
            given[opt] = argv[i + 1];
        }
        for (const auto& kv : given) apply(kv.first, kv.second);
        if (given.empty() && argc > 1) {
            inputFile_ = argv[1];
            if (argc > 2) {
                try { K_ = static_cast<size_t>(std::stoi(argv[2])); }
                catch (const std::exception& e) { std::cerr << "Invalid argument: " << e.what() << std::endl; }
            }
        }
    }

    std::string programPath() const { return programPath_; }
    std::string programName() const { return programName_; }
    std::string inputFile() const { return inputFile_; }
    size_t K() const { return K_; }
    int numIterations() const { return numIterations_; }
    float similarityThresholdAlpha() const { return similarityThresholdAlpha_; }
    float blockDensityThresholdDelta() const { return blockDensityThresholdDelta_; }
    bool testMode() const { return testMode_; }
    std::string outputLogDirectory() const { return outputLogDirectory_; }

private:
    void apply(const std::string& opt, const std::string& value) {
        if (opt.size() != 2) return;
        try {
            switch (opt[1]) {
            case 'f': case 'F': inputFile_ = value; break;
            case 'k': case 'K': K_ = static_cast<size_t>(std::stoi(value)); break;
            case 'a': case 'A': similarityThresholdAlpha_ = std::stof(value); break;
            case 'd': case 'D': blockDensityThresholdDelta_ = std::stof(value); break;
            case 't': case 'T': testMode_ = std::stoi(value) != 0; break;
            case 'l': case 'L': outputLogDirectory_ = value; break;
            default: break;
            }
        } catch (const std::invalid_argument& e) {
            std::cerr << "Invalid argument: " << e.what() << std::endl;
        } catch (const std::out_of_range& e) {
            std::cerr << "Out of range: " << e.what() << std::endl;
        }
    }

    std::string programPath_;
    std::string programName_;
    std::string inputFile_;
    std::string outputLogDirectory_;
    size_t K_ = 32;
    int numIterations_ = 10;
    float similarityThresholdAlpha_ = 0.3f;
    float blockDensityThresholdDelta_ = 0.3f;
    bool testMode_ = false;
};
