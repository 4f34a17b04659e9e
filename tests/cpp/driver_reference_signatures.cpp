// The flow of the reference's SfM drivers between "create Line3D object" and "save as txt" (main_vsfm.cpp:115-313, main_bundler.cpp:287),
// written against include/line3D_amd.hpp with the reference's OWN call shapes: addImage(id, image, K, R, t, worldpointIDs, max_width,
// loadAndStore) with a cv::Mat-shaped image and Eigen-shaped cameras (test doubles, ref_type_doubles.hpp).  The images are size stubs; the
// segments come from the segment caches in <data directory> (line3D.cc:143-168).
//   driver_reference_signatures <scene.nvm> <image folder> <output folder> <neighbors> <diffusion> [max_width=-1] [fixed_sim=0]
#include <cstdlib>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "ref_type_doubles.hpp"
#include "line3D_amd.hpp"

int main(int argc, char** argv)
{
    if (argc < 6) return 2;
    const std::string nvmFile = argv[1], inputFolder = argv[2], outputFolder = argv[3];
    const int neighbors = atoi(argv[4]);
    const bool diffusion = atoi(argv[5]) != 0;
    const int max_width = argc > 6 ? atoi(argv[6]) : -1;
    const bool fixed_sim = argc > 7 && atoi(argv[7]) != 0;
    const float max_uncertainty = 5.0f, min_uncertainty = 1.0f, sigma_p = 3.5f, sigma_a = 10.0f, min_baseline = 0.25f;
    const bool collinearity = true, verbose = true, loadAndStore = true;

    l3d_sfm_scene* scene = nullptr;
    if (l3d_sfm_read_nvm(nvmFile.c_str(), &scene) != L3D_OK) { fprintf(stderr, "%s\n", l3d_sfm_last_error(scene)); l3d_sfm_free(scene); return 1; }
    const unsigned int num_cams = (unsigned int)l3d_sfm_num_cameras(scene);
    std::vector<std::string> cams_imgFilenames(num_cams);
    std::vector<float> cams_focals(num_cams);
    std::vector<Eigen::Matrix3d> cams_rotation(num_cams);
    std::vector<Eigen::Vector3d> cams_translation(num_cams);
    std::vector<std::list<unsigned int> > cams_worldpointIDs(num_cams);
    for (unsigned int i = 0; i < num_cams; ++i) {
        double focal = 0, dist[2], R[9], t[3];
        int nwp = 0;
        l3d_sfm_camera(scene, (int)i, &focal, dist, R, t, &nwp);
        cams_imgFilenames[i] = l3d_sfm_camera_name(scene, (int)i);
        cams_focals[i] = (float)focal;
        for (int a = 0; a < 3; ++a) { cams_translation[i](a) = t[a]; for (int b = 0; b < 3; ++b) cams_rotation[i](a, b) = R[a * 3 + b]; }
        std::vector<uint32_t> ids((size_t)nwp);
        l3d_sfm_camera_worldpoints(scene, (int)i, ids.data());
        cams_worldpointIDs[i].assign(ids.begin(), ids.end());
    }
    l3d_sfm_free(scene);

    std::string data_directory = outputFolder + "/L3D_data/";
    L3D::Line3D* line3D = new L3D::Line3D(data_directory, neighbors, max_uncertainty, min_uncertainty, sigma_p, sigma_a, min_baseline, collinearity, verbose);
    if (!line3D->valid()) { delete line3D; return 1; }

    for (unsigned int i = 0; i < num_cams; ++i) {
        cv::Mat image = cv::imread(inputFolder + "/" + cams_imgFilenames[i]);
        float px = float(image.cols) / 2.0f;
        float py = float(image.rows) / 2.0f;
        float f = cams_focals[i];
        Eigen::Matrix3d K = Eigen::Matrix3d::Zero();
        K(0, 0) = f;
        K(1, 1) = f;
        K(0, 2) = px;
        K(1, 2) = py;
        K(2, 2) = 1.0;
        if (fixed_sim) {        // (the signature of line3D.h:75-79; every other view equally similar -- the neighbour choice is then by id)
            std::map<unsigned int, float> sim;
            for (unsigned int j = 0; j < num_cams; ++j) if (j != i) sim[j] = 1.0f / float(1 + (j > i ? j - i : i - j));
            line3D->addImage_fixed_sim(i, image, K, cams_rotation[i], cams_translation[i], sim, max_width, loadAndStore);
        } else
            line3D->addImage(i, image, K, cams_rotation[i], cams_translation[i], cams_worldpointIDs[i], max_width, loadAndStore);
    }
    // the guards of line3D.cc:101-127 through the same signature: an id in use, an empty image, an image without a cache -- printed, no view added
    {
        const unsigned int before = line3D->numCameras();
        cv::Mat image = cv::imread(inputFolder + "/" + cams_imgFilenames[0]), none;
        Eigen::Matrix3d K = Eigen::Matrix3d::Zero();
        K(0, 0) = K(1, 1) = K(2, 2) = 1.0;
        line3D->addImage(0, image, K, cams_rotation[0], cams_translation[0], cams_worldpointIDs[0], max_width, loadAndStore);
        line3D->addImage(num_cams + 1, none, K, cams_rotation[0], cams_translation[0], cams_worldpointIDs[0], max_width, loadAndStore);
        line3D->addImage(num_cams + 2, image, K, cams_rotation[0], cams_translation[0], cams_worldpointIDs[0], max_width, loadAndStore);
        if (line3D->numCameras() != before) { fprintf(stderr, "a guarded addImage added a view\n"); return 4; }
    }

    line3D->compute3Dmodel(diffusion);
    std::list<L3D::L3DFinalLine3D> result;
    line3D->getResult(result);

    std::stringstream str;
    str << "/line3D_result__";
    str << "W_" << max_width << "__";
    if (neighbors < 0) str << "N_ALL__"; else str << "N_" << neighbors << "__";
    str << "tL_" << min_uncertainty << "__";
    str << "tU_" << max_uncertainty << "__";
    str << "sigmaP_" << sigma_p << "__";
    str << "sigmaA_" << sigma_a << "__";
    str << (collinearity ? "COLLIN__" : "NO_COLLIN__");
    str << (diffusion ? "DIFFUSION" : "NO_DIFFUSION");
    line3D->save3DLinesAsSTL(result, outputFolder + str.str() + ".stl");
    line3D->save3DLinesAsTXT(result, outputFolder + str.str() + ".txt");

    unsigned int num_indiv_segments = 0;
    std::list<L3D::L3DFinalLine3D>::iterator rit = result.begin();
    for (; rit != result.end(); ++rit) {
        L3D::L3DFinalLine3D fl = *rit;
        num_indiv_segments += fl.segments3D()->size();
    }
    std::cout << "3D lines:        " << result.size() << std::endl;
    std::cout << "3D segments:     " << num_indiv_segments << std::endl;
    std::cout << "#images:         " << line3D->numCameras() << std::endl;
    delete line3D;
    return result.empty() ? 3 : 0;
}
